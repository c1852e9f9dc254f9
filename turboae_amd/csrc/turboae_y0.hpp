// Layout of Y0 - the layer-0 outputs of the f16x2 recurrent stacks (gru_rec_h<true> / gru_rec0u / rnn_rec_u<.., layer 0> write it,
// gru_l1f / rnn_l1f_u / gru_proj_h / rnn_proj_u read it as the layer-1 projection's operand) - in HBM.
//
// Logically a row is one (position, block): 200 halves hi | 200 halves lo in K order [forward units 0..99 | backward units 0..99],
// read by the consumers as 25 sixteen-byte pieces per plane (piece p = halves 8p .. 8p + 7).  Through r05 the rows lay in HBM as they
// are read (16 rows x 800 bytes per step); but the two directions are different workgroups that reach step t at different times, so
// every 128-byte line of such a row was written half by one and - microseconds to milliseconds later - half by the other (timing
// build: -3 % of the GRU decoder's forward with direction-private rows, LABNOTES 12.4).  Since late r06 the 12 800 bytes of a
// (16-block group, step) are three regions, K order and piece contents unchanged:
//     A  [plane][row n][pieces 0..11]    2 x 16 x 192 B   forward units 0..95           written by the forward wave only
//     B  [plane][row n][pieces 13..24]   2 x 16 x 192 B   backward units 4..99          written by the backward wave only
//     C  [plane][row n][piece 12]        2 x 16 x  16 B   forward 96..99 | backward 0..3   the one piece both directions share
// A consumer's piece p of row n is at y0_piece(n, p) (+ y0_lo_add(p) for the lo plane); pieces past 24 (the K padding of the last
// slab, multiplied by zero weights) read on into initialised halves of the same step as before.
#pragma once
#include <stdint.h>

namespace tae {

constexpr uint32_t kY0StepB = 16 * 800;                   // bytes per (16-block group, step): unchanged
constexpr uint32_t kY0A = 0, kY0B = 6144, kY0C = 12288;   // region bases inside a step
constexpr uint32_t kY0PlaneAB = 3072, kY0PlaneC = 256;    // hi -> lo inside a region

__host__ __device__ __forceinline__ constexpr uint32_t y0_piece(int n, int p) {
    return p < 12 ? kY0A + (uint32_t)n * 192u + (uint32_t)p * 16u
                  : (p == 12 ? kY0C + (uint32_t)n * 16u : kY0B + (uint32_t)n * 192u + (uint32_t)(p - 13) * 16u);
}
__host__ __device__ __forceinline__ constexpr uint32_t y0_lo_add(int p) { return p == 12 ? kY0PlaneC : kY0PlaneAB; }
// byte offset of logical half j (0..199) of row n, hi plane, and of its lo twin
__host__ __device__ __forceinline__ constexpr uint32_t y0_half(int n, int j) { return y0_piece(n, j >> 3) + (uint32_t)(j & 7) * 2u; }
__host__ __device__ __forceinline__ constexpr uint32_t y0_half_lo(int n, int j) { return y0_half(n, j) + y0_lo_add(j >> 3); }

// Writers: lane (n, q) of the wave that owns unit tile u of direction dir stores units 16u + 4q .. + 3 (8 bytes per plane).
// For u >= 1 the offsets are linear in u (+ 32 bytes per tile, lo = hi + kY0PlaneAB); tile 0 of the backward direction has the one
// lane (q = 0) whose units 0..3 live in region C.
struct Y0UnitOff { uint32_t hi0, lo0, hi1; };              // tile 0: hi0 / lo0; tile u >= 1: hi1 + (u - 1) * 32, lo = that + kY0PlaneAB
__device__ __forceinline__ Y0UnitOff y0_unit_offsets(int dir, int n, int q) {
    Y0UnitOff o;
    o.hi0 = y0_half(n, dir * 100 + 4 * q);
    o.lo0 = y0_half_lo(n, dir * 100 + 4 * q);
    o.hi1 = y0_half(n, dir * 100 + 16 + 4 * q);
    return o;
}
__device__ __forceinline__ void y0_unit_tile(int dir, int u, int n, int q, uint32_t& hi, uint32_t& lo) {      // one tile (u wave-uniform)
    hi = y0_half(n, dir * 100 + 16 * u + 4 * q);
    lo = y0_half_lo(n, dir * 100 + 16 * u + 4 * q);
}

}  // namespace tae
