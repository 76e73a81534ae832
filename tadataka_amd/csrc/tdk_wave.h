// tdk_wave.h -- wave64 register-level reductions for gfx950 (v_permlane{32,16}_swap + DPP),
// shared by the DVO evaluation and the bundle-adjustment block reduce.
#pragma once

#include <hip/hip_runtime.h>

namespace tdk {

// a[lanes 32..63] <-> b[lanes 0..31] (v_permlane32_swap, gfx950)
__device__ __forceinline__ void swap_halves(double &a, double &b) {
    auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    a = __hiloint2double((int)hi[0], (int)lo[0]);
    b = __hiloint2double((int)hi[1], (int)lo[1]);
}

// a[rows 1, 3] <-> b[rows 0, 2] (v_permlane16_swap, rows of 16 lanes)
__device__ __forceinline__ void swap_rows(double &a, double &b) {
    auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    a = __hiloint2double((int)hi[0], (int)lo[0]);
    b = __hiloint2double((int)hi[1], (int)lo[1]);
}

template <int CTRL>
__device__ __forceinline__ double dpp_move(double x) {
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// Wave64 sum of up to 32 accumulators at once.  Every halving step folds two
// accumulators into one register (each half of the lanes keeps one of them and
// hands the other to its partner), so 32 -> 16 -> 8 -> 4 -> 2 -> 1 registers
// take 31 exchanges instead of 30 * 6: lanes 2j and 2j+1 end up holding the
// wave total of accumulator j.  Cross-row steps are v_permlane{32,16}_swap,
// in-row steps are DPP mirrors; nothing goes through LDS.
template <int N>
__device__ __forceinline__ double wave_sum_transposed(const double (&acc)[N]) {
    static_assert(N <= 32, "at most 32 accumulators per wave");
    const int lane = threadIdx.x & 63;
    double v[32];
#pragma unroll
    for (int k = 0; k < 32; k++) v[k] = k < N ? acc[k] : 0.0;
#pragma unroll
    for (int k = 0; k < 16; k++) {   // lane bit 5 selects accumulator bit 4
        swap_halves(v[k], v[k + 16]);
        v[k] += v[k + 16];
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {    // lane bit 4 -> bit 3
        swap_rows(v[k], v[k + 8]);
        v[k] += v[k + 8];
    }
    const bool b3 = (lane & 8) != 0, b2 = (lane & 4) != 0, b1 = (lane & 2) != 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {    // lane i <-> 15 - i within a row: bit 3 -> bit 2
        const double keep = b3 ? v[k + 4] : v[k], send = b3 ? v[k] : v[k + 4];
        v[k] = keep + dpp_move<0x140>(send);   // row_mirror
    }
#pragma unroll
    for (int k = 0; k < 2; k++) {    // i <-> 7 - i within 8 lanes: bit 2 -> bit 1
        const double keep = b2 ? v[k + 2] : v[k], send = b2 ? v[k] : v[k + 2];
        v[k] = keep + dpp_move<0x141>(send);   // row_half_mirror
    }
    {                                // i <-> 3 - i within a quad: bit 1 -> bit 0
        const double keep = b1 ? v[1] : v[0], send = b1 ? v[0] : v[1];
        v[0] = keep + dpp_move<0x1B>(send);    // quad_perm [3,2,1,0]
    }
    return v[0] + dpp_move<0xB1>(v[0]);        // quad_perm [1,0,3,2]
}

}  // namespace tdk
