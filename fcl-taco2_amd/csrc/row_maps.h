// row_maps.h -- the device-built row / frame maps (fcl_row_maps_build) as workgroup-level device functions, shared by the standalone kernels
// (pointwise.hip) and by the extra workgroup that builds them inside the BiLSTM launch (bilstm.hip: forced durations do not depend on the encoder,
// so the maps ride beside the recurrence instead of costing two launches on the pass's dependent chain).
#pragma once
#include "fcl_common.h"

namespace fcl {

constexpr int RM_CHUNK = 4096;
constexpr int RM_DMAX = 65535;  // durations are clamped here (sum over <= 32 k rows stays inside int32); anything near it trips FCL_STATUS_LMAX_CAP

__device__ __forceinline__ int rm_dur(const fcl_row_maps_t& a, int j) {
    const long long d = a.dur_i32 ? (long long)a.dur_i32[j] : (long long)a.dur_i64[a.row_src ? a.row_src[j] : j];
    const int v = (int)(d < 0 ? 0 : (d > RM_DMAX ? RM_DMAX : d));
    return (a.pad && a.pad[j]) ? -1 - v : v;  // padding rows: tagged negative for the zero-duration test, counted as 0 everywhere else
}

constexpr int RMF_V = 256;  // value buckets 0 .. 254; 255 = anything larger (then > lmax_cap: the pass is void anyway)

// NT threads (a multiple of 64, >= 320): the standalone launch uses 1 024, the workgroup that rides along in the BiLSTM launch 512.
template <int NT>
__device__ __forceinline__ void row_maps_block(const fcl_row_maps_t& a) {
    constexpr int NW = NT / 64;
    static_assert(NT % 64 == 0 && NT > RMF_V, "row_maps_block: thread RMF_V carries the duration prefix");
    __shared__ unsigned short cnt[NW][RMF_V];
    __shared__ int running[RMF_V], start[RMF_V], wsum[NW], carry, zeros_l, dmax_l;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int k = tid; k < RMF_V; k += NT) running[k] = 0;
    if (tid == 0) { carry = 0; zeros_l = 0; dmax_l = 0; }
    int my_zero = 0, my_max = 0;
    const int rounds = (a.n + NT - 1) / NT;
    for (int r = 0; r < rounds; ++r) {
        const int i = r * NT + tid;
        const bool valid = i < a.n;
        const int raw = valid ? rm_dur(a, i) : -1;  // padding rows are tagged negative
        const int d = max(raw, 0), v = min(d, RMF_V - 1);
        my_zero += valid && raw == 0;
        my_max = max(my_max, d);
        __syncthreads();  // (the previous round is done with cnt / wsum)
        for (int k = tid; k < NW * RMF_V; k += NT) (&cnt[0][0])[k] = 0;
        __syncthreads();
        // stable rank inside the (wave, value) group: lanes below me with my value
        int inw = 0;
        unsigned long long todo = __ballot(valid);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int lv = __shfl(v, leader);
            const unsigned long long mask = __ballot(valid && v == lv);
            if (valid && v == lv) inw = __popcll(mask & ((1ull << lane) - 1ull));
            if (lane == leader) cnt[wave][lv] = (unsigned short)__popcll(mask);
            todo &= ~mask;
        }
        // exclusive prefix sum of the durations inside the wave
        int incl = d;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        if (tid < RMF_V) {  // rows of value `tid` before each wave of this round, earlier rounds included (<= N <= 32 768: fits 16 bits)
            int s = running[tid];
            for (int w = 0; w < NW; ++w) {
                const int c = cnt[w][tid];
                cnt[w][tid] = (unsigned short)s;
                s += c;
            }
            running[tid] = s;
        } else if (tid == RMF_V) {
            int s = carry;
            for (int w = 0; w < NW; ++w) {
                const int t = wsum[w];
                wsum[w] = s;
                s += t;
            }
            carry = s;
        }
        __syncthreads();
        if (valid) {
            a.scratch[i] = (int)cnt[wave][v] + inw;          // rows with my value before me
            a.scratch[a.n + i] = wsum[wave] + incl - d;      // frames before me = my first output frame (H10)
        }
    }
    atomicAdd(&zeros_l, my_zero);
    atomicMax(&dmax_l, my_max);
    __syncthreads();
    if (tid == 0) {  // rows with a larger duration, per value (suffix sums of the 256 totals)
        int s = 0;
        for (int v = RMF_V - 1; v >= 0; --v) {
            start[v] = s;
            s += running[v];
        }
    }
    __syncthreads();
    __threadfence_block();
    for (int r = 0; r < rounds; ++r) {
        const int i = r * NT + tid;
        if (i >= a.n) break;
        const int d = max(rm_dur(a, i), 0), v = min(d, RMF_V - 1);
        const int rank = start[v] + a.scratch[i];
        a.src_rows[rank] = a.row_src ? a.row_src[i] : i;
        a.dur_sorted[rank] = d;
        a.frame_off[rank] = a.scratch[a.n + i];
        if (a.order) a.order[rank] = i;
    }
    const int total = carry;
    for (int t = tid; t <= a.lmax_cap; t += NT) a.live_rows[t] = t < RMF_V ? start[t] : 0;
    for (int b = tid; b <= a.b; b += NT) {
        const int row0 = a.utt_row0 ? a.utt_row0[b] : b * a.t_max;
        a.utt_frame0[b] = row0 < a.n ? a.scratch[a.n + row0] : total;
    }
    if (tid == 0) {
        a.totals[0] = total;
        a.totals[1] = dmax_l;
        a.totals[2] = zeros_l;
        a.totals[3] = 0;
        unsigned int bits = 0;
        if (zeros_l) bits |= FCL_STATUS_ZERO_DURATION;
        if (dmax_l > a.lmax_cap) bits |= FCL_STATUS_LMAX_CAP;
        if (total > a.frames_cap) bits |= FCL_STATUS_FRAMES_CAP;
        if (bits) atomicOr(a.status, bits);
    }
}

// second launch (the totals are complete): frame -> utterance bounds, and -- on any violation -- no live rows at all, so that a decoder loop
// driven by these maps neither runs past its launched steps nor scatters past the frame buffers
// frames [f0, f0 + count) handled by the caller's threads one each (the grid form) or in a strided loop (row_maps_finish_block)
__device__ __forceinline__ void row_maps_finish_item(const fcl_row_maps_t& a, int f) {
    const int total = a.totals[0];
    const bool bad = a.totals[2] > 0 || a.totals[1] > a.lmax_cap || total > a.frames_cap;
    if (bad && f <= a.lmax_cap) a.live_rows[f] = 0;
    if (f >= a.frames_cap) return;
    int lo = 0, hi = 0;
    if (!bad && f < total) {
        int l = 0, r = a.b;  // utt_frame0[l] <= f < utt_frame0[r]
        while (r - l > 1) {
            const int mid = (l + r) >> 1;
            if (a.utt_frame0[mid] <= f) l = mid; else r = mid;
        }
        lo = a.utt_frame0[l];
        hi = a.utt_frame0[l + 1];
    }
    a.frame_lo[f] = lo;
    a.frame_hi[f] = hi;
}


// the finish step by ONE workgroup of NT threads, after row_maps_block<NT> of the same workgroup (its writes are visible after the barrier)
template <int NT>
__device__ __forceinline__ void row_maps_finish_block(const fcl_row_maps_t& a) {
    __threadfence_block();
    __syncthreads();
    const int items = max(a.frames_cap, a.lmax_cap + 1);
    for (int f = threadIdx.x; f < items; f += NT) row_maps_finish_item(a, f);
}

}  // namespace fcl
