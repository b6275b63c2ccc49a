#!/usr/bin/env python3
"""Condenses the CSV summaries of one `collect.sh` run into the JSON that bench.py reads for `roofline.traffic`,
`roofline.valu_issue` and the matrix-core utilisation of the grid scoring.

    python profiles/make_counters_json.py <dir with kernel_stats.csv, pmc_*.csv> <tag> <out.json> [hypotheses per launch, default 512]

Per-launch values: a --pmc CSV row holds the sum over all dispatch rows of a kernel (rocprofv3 writes one row per
launch and counter instance), so a per-launch value is Total / launches, launches taken from the FETCH_SIZE pass
(one row per launch).  The shader clock is SQ_BUSY_CYCLES per instance / the kernel's average duration.
"""
import csv
import json
import sys

COMPUTE_UNITS = 256


def rows(path):
    with open(path) as fh:
        return list(csv.DictReader(fh))


def main(d, tag, out, n_hyp=512):
    n_hyp = int(n_hyp)
    stats = {r["Name"]: r for r in rows(f"{d}/kernel_stats.csv")}
    pmc = {}
    for name in ("fetch_size", "write_size", "sq", "mfma"):
        try:
            for r in rows(f"{d}/pmc_{name}.csv"):
                pmc[(r["Kernel"], r["Counter"])] = r
        except FileNotFoundError:
            pass

    def kern(sub):
        names = [k for k in stats if sub in k]
        return max(names, key=lambda k: float(stats[k]["TotalDurationNs"]))

    def per_launch(k, counter):
        r = pmc.get((k, counter))
        if r is None:
            return None
        launches = int(pmc[(k, "FETCH_SIZE")]["Dispatches"])
        return float(r["Total"]) / launches

    def clock_ghz(k):
        r = pmc[(k, "SQ_BUSY_CYCLES")]
        return float(r["AveragePerDispatch"]) / float(stats[k]["AverageNs"])

    solve, mfma = kern("solve_kernel"), kern("grid_mfma_kernel")

    def helper(sub, exclude=None):
        """A launch that belongs to the solve call besides the loop (the launch that lists the first step's tiles, the
        kernel that computes them): duration and counters per launch, or None when the run had none."""
        names = [k for k in stats if sub in k and k != exclude]
        if not names:
            return None
        k = min(names, key=lambda n: float(stats[n]["TotalDurationNs"])) if sub == "solve_kernel" else names[0]
        return {"kernel": k, "avg_duration_ms": float(stats[k]["AverageNs"]) / 1e6,
                "fetch_size_kb_per_launch": per_launch(k, "FETCH_SIZE"), "write_size_kb_per_launch": per_launch(k, "WRITE_SIZE"),
                "sq_insts_valu_per_launch": per_launch(k, "SQ_INSTS_VALU")}
    res = {
        "tag": tag,
        "command": "bash profiles/collect.sh (rocprofv3 --kernel-trace --stats; --pmc FETCH_SIZE | WRITE_SIZE | SQ_* | MFMA counters, one run each, "
                   "on python3 bench.py --no-cpu-baseline --streams 1)",
        "kernel": solve, "hypotheses_per_launch": n_hyp, "adam_steps": 300, "on_demand_table": "ILb1ELb0ELb1E" in solve or "ILb0ELb0ELb1E" in solve,
        "compute_units": COMPUTE_UNITS,
        "avg_duration_ms": float(stats[solve]["AverageNs"]) / 1e6,
        "shader_clock_ghz": round(clock_ghz(solve), 3),
        "fetch_size_kb_per_launch": per_launch(solve, "FETCH_SIZE"),
        "write_size_kb_per_launch": per_launch(solve, "WRITE_SIZE"),
        "sq_insts_valu_per_launch": per_launch(solve, "SQ_INSTS_VALU"),
        "sq_insts_salu_per_launch": per_launch(solve, "SQ_INSTS_SALU"),
        "sq_insts_lds_per_launch": per_launch(solve, "SQ_INSTS_LDS"),
        "sq_wave_quad_cycles_per_launch": per_launch(solve, "SQ_WAVE_CYCLES"),
        "sq_active_inst_valu_quad_cycles_per_launch": per_launch(solve, "SQ_ACTIVE_INST_VALU"),
        "sq_wait_inst_any_quad_cycles_per_launch": per_launch(solve, "SQ_WAIT_INST_ANY"),
        "sq_insts_mfma_per_launch": per_launch(solve, "SQ_INSTS_MFMA"),
        "note": "FETCH_SIZE / WRITE_SIZE are KB of fabric-side L2 requests (Infinity-Cache hits included); the gfx950 x2 correction of FETCH_SIZE "
                "applies to wide streaming reads, this kernel mostly gathers single 16-byte table elements, so the raw value is kept. The writes "
                "are the computed table tiles, the register spills (scratch) and the results.",
        "footprint": helper("solve_kernel", exclude=solve), "tile_bulk": helper("tile_bulk_kernel"),
        "grid_mfma": {
            "kernel": mfma, "images_per_launch": n_hyp, "avg_duration_ms": float(stats[mfma]["AverageNs"]) / 1e6,
            "shader_clock_ghz": round(clock_ghz(mfma), 3),
            "sq_insts_mfma_per_launch": per_launch(mfma, "SQ_INSTS_MFMA"),
            "sq_valu_mfma_busy_cycles_per_launch": per_launch(mfma, "SQ_VALU_MFMA_BUSY_CYCLES"),
            "sq_insts_valu_mfma_mops_f16_per_launch": per_launch(mfma, "SQ_INSTS_VALU_MFMA_MOPS_F16"),
            "fetch_size_kb_per_launch": per_launch(mfma, "FETCH_SIZE"),
        },
    }
    g = res["grid_mfma"]
    if g["sq_valu_mfma_busy_cycles_per_launch"]:
        # matrix-pipe utilisation: busy cycles summed over the SIMDs / (SIMDs x cycles of the launch)
        g["mfma_busy_frac"] = g["sq_valu_mfma_busy_cycles_per_launch"] / (COMPUTE_UNITS * 4 * g["avg_duration_ms"] * 1e-3 * g["shader_clock_ghz"] * 1e9)
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:5])
