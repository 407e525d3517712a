// tsq_wavescan.h — segmented inclusive scans over the 64 lanes of a wave (device code).
//
// Lanes hold CONSECUTIVE items; a run is a maximal stretch of lanes that belong together (equal group slot in tsq_streamagg.h, equal
// LDS cell in tsq_daagg.h); `head` marks the first lane of a run.  After the scan the LAST lane of every run holds the run's total, so
// one atomic per (wave, run) replaces one per item — what makes a hot key cheap: 64 lanes updating the SAME LDS or HBM word are 64
// serialised atomics.  The combine condition of step k depends on the run boundaries only, so it is computed once (a 6-bit mask per
// lane) and every value of the item is scanned under it.
#ifndef TSQ_WAVESCAN_H
#define TSQ_WAVESCAN_H

#include <hip/hip_runtime.h>

// step k combines lane l with lane l - 2^k iff bit k of the result is set
__device__ __forceinline__ uint32_t sa_cond_mask(bool head, uint32_t lane) {
    uint32_t cm = 0;
    bool f = head;
    for (int k = 0; k < 6; k++) {
        const int o = 1 << k;
        const bool fu = __shfl_up((int)f, o, 64) != 0;
        if (lane >= (uint32_t)o && !f) {
            cm |= 1u << k;
            f = fu;
        }
    }
    return cm;
}
__device__ __forceinline__ uint64_t sa_scan_add(uint64_t v, uint32_t cm) {
    for (int k = 0; k < 6; k++) {
        const uint64_t y = __shfl_up(v, 1 << k, 64);
        if ((cm >> k) & 1u) v += y;
    }
    return v;
}
__device__ __forceinline__ double sa_scan_addf(double v, uint32_t cm) {
    for (int k = 0; k < 6; k++) {
        const double y = __shfl_up(v, 1 << k, 64);
        if ((cm >> k) & 1u) v += y;
    }
    return v;
}
__device__ __forceinline__ uint64_t sa_scan_max(uint64_t v, uint32_t cm) {
    for (int k = 0; k < 6; k++) {
        const uint64_t y = __shfl_up(v, 1 << k, 64);
        if (((cm >> k) & 1u) && y > v) v = y;
    }
    return v;
}
__device__ __forceinline__ uint64_t sa_scan_min(uint64_t v, uint32_t cm) {
    for (int k = 0; k < 6; k++) {
        const uint64_t y = __shfl_up(v, 1 << k, 64);
        if (((cm >> k) & 1u) && y < v) v = y;
    }
    return v;
}
// 128-bit sums of int64 addends: (lo, hi) with hi = the signed high word
__device__ __forceinline__ void sa_scan_add128(uint64_t& lo, int64_t& hi, uint32_t cm) {
    for (int k = 0; k < 6; k++) {
        const uint64_t ylo = __shfl_up(lo, 1 << k, 64);
        const int64_t yhi = (int64_t)__shfl_up((uint64_t)hi, 1 << k, 64);
        if ((cm >> k) & 1u) {
            const uint64_t s = lo + ylo;
            hi += yhi + (s < lo ? 1 : 0);
            lo = s;
        }
    }
}

#endif
