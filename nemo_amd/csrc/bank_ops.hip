// Feature-bank bookkeeping of multi-object training batches (the reference's CuNeMo extension,
// cu_layers/CuNeMo/csrc/gather_features/gather_features.cu and mask_weight/mask_weight.cu).
//
// An image holds up to M objects; its K (padded) vertex slots are the concatenation of the objects'
// vertices in the order of sample_indexs[n, :] (mesh index per object, -1 = no object), object i starting
// at slot sample_shifts[n, i].  The bank holds the vertices of all L meshes back to back, mesh l starting
// at bank_shifts[l].
//
//   slot_map        (n, k) -> bank vertex (or -1 for padding slots), the search loop of the reference
//   gather          bank vertex v <- sum over the slots that map to it of weight * feature.  The reference
//                   scatters with one atomicAdd per channel; here a workgroup owns one bank vertex and walks
//                   the (image, object) pairs that hold its mesh in a fixed order: no atomics, no zero-fill
//                   pass, and the sums are reproducible.
//   gather backward one wave per slot: a slot has exactly one bank vertex, so nothing needs an atomic.
//   gather_idx / mask_weight   per-slot object bookkeeping and the per-(slot, bank vertex) loss weight.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nemo {

// The object a slot belongs to: the last object whose first slot is <= k (gather_features.cu:48-60,
// mask_weight.cu:36-49).  Returns the object's position i in the image's list, or -1 when the list starts
// after k.
__device__ inline int slot_object(const int32_t* __restrict__ shifts, int M, int k) {
    int found = -1;
    for (int i = 0; i < M; ++i) {
        const int next = shifts[i];
        if (next == -1 || next > k) break;
        found = i;
    }
    return found;
}

// forward_mapping of the reference: (n, k) -> bank vertex, -1 for k >= input_valid[n].
__global__ __launch_bounds__(256) void slot_map_kernel(const int32_t* __restrict__ input_valid, const int32_t* __restrict__ bank_shifts,
                                                       const int32_t* __restrict__ sample_shifts,
                                                       const int32_t* __restrict__ sample_indexs, int N, int M, int K,
                                                       int32_t* __restrict__ forward_mapping) {
    const int pid = blockIdx.x * blockDim.x + threadIdx.x;
    if (pid >= N * K) return;
    const int n = pid / K, k = pid - n * K;
    int out = -1;
    if (k < input_valid[n]) {
        const int i = slot_object(sample_shifts + (size_t)n * M, M, k);
        // a list that starts after slot k leaves mesh 0 at shift 0 in the reference's loop
        const int mesh = i >= 0 ? sample_indexs[(size_t)n * M + i] : 0;
        const int shift = i >= 0 ? sample_shifts[(size_t)n * M + i] : 0;
        if (mesh >= 0) out = k - shift + bank_shifts[mesh];      // an empty entry of the list owns no slot
    }
    forward_mapping[pid] = out;
}

// One workgroup per bank vertex v (mesh l, vertex j of that mesh); threads over channels.  Every (image,
// object) pair that holds mesh l offers slot k = shift + j; the slot counts when the map sends it to v
// (it does whenever the shifts are the running sums of the objects' vertex counts, as the callers build them).
__global__ __launch_bounds__(128) void gather_features_kernel(const float* __restrict__ feat, const float* __restrict__ weight,
                                                              const int32_t* __restrict__ bank_shifts,
                                                              const int32_t* __restrict__ sample_shifts,
                                                              const int32_t* __restrict__ sample_indexs,
                                                              const int32_t* __restrict__ forward_mapping, int N, int M, int K,
                                                              int C, int L, float* __restrict__ out_feat,
                                                              float* __restrict__ out_weight) {
    const int v = blockIdx.x;
    int l = 0;
    for (int q = 1; q < L; ++q)
        if (bank_shifts[q] <= v) l = q;          // last mesh that starts at or before v (empty meshes share a start)
    const int j = v - bank_shifts[l];
    float wsum = 0.0f;
    for (int c0 = 0; c0 < C; c0 += blockDim.x) {
        const int c = c0 + threadIdx.x;
        float acc = 0.0f;
        wsum = 0.0f;
        for (int n = 0; n < N; ++n)
            for (int i = 0; i < M; ++i) {
                if (sample_indexs[(size_t)n * M + i] != l) continue;
                const int k = sample_shifts[(size_t)n * M + i] + j;
                if (k < 0 || k >= K || forward_mapping[(size_t)n * K + k] != v) continue;
                const float w = weight[(size_t)n * K + k];
                wsum += w;
                if (c < C) acc += feat[((size_t)n * K + k) * C + c] * w;
            }
        if (c < C) out_feat[(size_t)v * C + c] = acc;
    }
    if (threadIdx.x == 0) out_weight[v] = wsum;
}

// One wave per slot: grad_weight = g_w[v] + <g_feat[v], feat>, grad_feat = weight * g_feat[v]
// (gather_features.cu:71-98); padding slots get zeros.
__global__ __launch_bounds__(256) void gather_features_backward_kernel(const float* __restrict__ g_feat, const float* __restrict__ g_weight,
                                                                       const float* __restrict__ feat,
                                                                       const float* __restrict__ weight,
                                                                       const int32_t* __restrict__ forward_mapping, int NK, int C,
                                                                       float* __restrict__ gi_feat, float* __restrict__ gi_weight) {
    const int lane = threadIdx.x & 63;
    const int pid = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (pid >= NK) return;
    const int v = forward_mapping[pid];
    if (v < 0) {
        for (int c = lane; c < C; c += 64) gi_feat[(size_t)pid * C + c] = 0.0f;
        if (lane == 0) gi_weight[pid] = 0.0f;
        return;
    }
    const float w = weight[pid];
    float dot = 0.0f;
    for (int c = lane; c < C; c += 64) {
        const float g = g_feat[(size_t)v * C + c];
        dot += g * feat[(size_t)pid * C + c];
        gi_feat[(size_t)pid * C + c] = g * w;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
    if (lane == 0) gi_weight[pid] = g_weight[v] + dot;
}

// Per slot: which object (mesh index), which vertex of it, where that mesh starts in the bank, how many
// vertices it has (mask_weight.cu:10-56); zeros for padding slots.
__global__ __launch_bounds__(256) void gather_idx_kernel(const int32_t* __restrict__ input_valid, const int32_t* __restrict__ sample_shifts,
                                                         const int32_t* __restrict__ sample_indexs,
                                                         const int32_t* __restrict__ sample_n_verts,
                                                         const int32_t* __restrict__ bank_starts, int N, int M, int K,
                                                         int32_t* __restrict__ object_idx, int32_t* __restrict__ source_vert_idx,
                                                         int32_t* __restrict__ verts_start, int32_t* __restrict__ num_verts) {
    const int pid = blockIdx.x * blockDim.x + threadIdx.x;
    if (pid >= N * K) return;
    const int n = pid / K, k = pid - n * K;
    int mesh = 0, src = 0, start = 0, nv = 0;
    if (k < input_valid[n]) {
        const int i = slot_object(sample_shifts + (size_t)n * M, M, k);
        if (i >= 0) { mesh = sample_indexs[(size_t)n * M + i]; nv = sample_n_verts[(size_t)n * M + i]; }
        src = k - (i >= 0 ? sample_shifts[(size_t)n * M + i] : 0);
        start = mesh >= 0 ? bank_starts[mesh] : 0;
    }
    object_idx[pid] = mesh; source_vert_idx[pid] = src; verts_start[pid] = start; num_verts[pid] = nv;
}

// out[s, k]: bank vertex k seen from the slot s - inside the slot's own mesh the distance-based weight
// of (own vertex, that vertex), in the clutter tail weight_c, elsewhere (other meshes) weight_x
// (mask_weight.cu:59-90).
__global__ __launch_bounds__(256) void mask_weight_kernel(const int32_t* __restrict__ object_idx, const int32_t* __restrict__ source_vert_idx,
                                                          const int32_t* __restrict__ verts_start,
                                                          const int32_t* __restrict__ verts_end,
                                                          const float* __restrict__ verts_dist_weight, float weight_x,
                                                          float weight_c, int clutter_start, int S, int K, int V,
                                                          float* __restrict__ out) {
    const int s = blockIdx.x;
    const int vs = verts_start[s], ve = verts_end[s];
    const float* row = verts_dist_weight + ((size_t)object_idx[s] * V + source_vert_idx[s]) * V;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        float val;
        if (k >= vs && k < ve) val = row[k - vs];
        else if (k >= clutter_start) val = weight_c;
        else val = weight_x;
        out[(size_t)s * K + k] = val;
    }
}

void launch_slot_map(const int32_t* input_valid, const int32_t* bank_shifts, const int32_t* sample_shifts, const int32_t* sample_indexs,
                     int N, int M, int K, int32_t* forward_mapping, hipStream_t st) {
    hipLaunchKernelGGL(slot_map_kernel, dim3((N * K + 255) / 256), dim3(256), 0, st, input_valid, bank_shifts, sample_shifts,
                       sample_indexs, N, M, K, forward_mapping);
}

void launch_gather_features(const float* feat, const float* weight, const int32_t* bank_shifts, const int32_t* sample_shifts,
                            const int32_t* sample_indexs, const int32_t* forward_mapping, int N, int M, int K, int C, int L, int KK,
                            float* out_feat, float* out_weight, hipStream_t st) {
    hipLaunchKernelGGL(gather_features_kernel, dim3(KK), dim3(128), 0, st, feat, weight, bank_shifts, sample_shifts, sample_indexs,
                       forward_mapping, N, M, K, C, L, out_feat, out_weight);
}

void launch_gather_features_backward(const float* g_feat, const float* g_weight, const float* feat, const float* weight,
                                     const int32_t* forward_mapping, int NK, int C, float* gi_feat, float* gi_weight, hipStream_t st) {
    hipLaunchKernelGGL(gather_features_backward_kernel, dim3((NK + 3) / 4), dim3(256), 0, st, g_feat, g_weight, feat, weight,
                       forward_mapping, NK, C, gi_feat, gi_weight);
}

void launch_gather_idx(const int32_t* input_valid, const int32_t* sample_shifts, const int32_t* sample_indexs,
                       const int32_t* sample_n_verts, const int32_t* bank_starts, int N, int M, int K, int32_t* object_idx,
                       int32_t* source_vert_idx, int32_t* verts_start, int32_t* num_verts, hipStream_t st) {
    hipLaunchKernelGGL(gather_idx_kernel, dim3((N * K + 255) / 256), dim3(256), 0, st, input_valid, sample_shifts, sample_indexs,
                       sample_n_verts, bank_starts, N, M, K, object_idx, source_vert_idx, verts_start, num_verts);
}

void launch_mask_weight(const int32_t* object_idx, const int32_t* source_vert_idx, const int32_t* verts_start, const int32_t* verts_end,
                        const float* verts_dist_weight, float weight_x, float weight_c, int clutter_start, int S, int K, int V,
                        float* out, hipStream_t st) {
    hipLaunchKernelGGL(mask_weight_kernel, dim3(S), dim3(256), 0, st, object_idx, source_vert_idx, verts_start,
                       verts_end, verts_dist_weight, weight_x, weight_c, clutter_start, S, K, V, out);
}

}  // namespace nemo
