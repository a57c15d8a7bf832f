// svdd_spt.h — how many whole sequences a backbone workgroup takes when several fit its 208-row tile (L <= 104).
// A tile costs its row tiles of 16 plus a fixed part (the loop skeleton of the 756 (layer, chunk, tap) iterations, LayerNorm
// phases, barriers, the first-layer lookup): measured with 256 x s sequences of L = 50 / 33 (tools/backbone_spt_calib.py,
// one full round): fp32 (round 5, interleaved sequences) 806 / 1058 / 1359 / 1659 us at 4 / 7 / 10 / 13 row tiles = 95 us per row tile +
// 4.5 row tiles' worth fixed (rounds 2-4, stacked sequences: 865 / 1240 / 1582 / 1941 = 120 us + 3); f16x3 (interleaved, on the
// transposed-accumulator kernel) 421 / 442 / 481 / 536 us = 13 us per row tile + 29 row tiles' worth fixed (stacked, round-2 kernel:
// 478 / 575 / 644 / 726 = 28 us + 13). The caller passes the fixed part in HALF row tiles (`fixed_half`: 9 fp32, 58 x3 modes, 90 one-pass).
// The launch costs rounds x tile cost, rounds = ceil(tiles / CUs) (one workgroup per CU). Packing the tile full is only
// best when the tiles then fill whole rounds: 256 RNA sequences (L = 50) packed four to a tile are 64 workgroups on 256
// CUs, one to a tile they are 256 workgroups of a quarter of the work; 1408 live candidates are 352 full tiles = 2
// rounds x 13 row tiles, but 470 tiles of three = 2 rounds x 10. Evaluated on the host when the row count is known
// there and on the device (every workgroup computes the same value) when it is a device scalar (work-skipping).
// A row's result does not depend on the choice (tests/test_fused_gpu.py::test_one_launch_backbone_vs_plain).
#pragma once

#ifndef SVDD_TILE_ROWS
#define SVDD_TILE_ROWS 208
#endif

// Cost of one tile of `rt` row tiles in HALF row tiles. Round 6: the fp32 kernel (fixed_half == 9) runs tiles of 5 .. 8 row tiles on
// a four-slot entry body and skips the LayerNorm passes of slots nobody owns (backbone_kernel, "Short tiles"): measured 817 / 994 /
// 1356 / 1667 us at 4 / 7 / 10 / 13 row tiles (L = 50) and 646 / 723 / 919 / 1164 / 1370 / 1559 us at 3 / 5 / 7 / 9 / 11 / 13 (L = 33), i.e.
// ~95 us per row tile + a fixed part worth 4.5 row tiles on the 7-slot and the 2-slot bodies, ~3 on the 4-slot body.
__host__ __device__ inline long long svdd_tile_cost(int rt, int fixed_half) {
  const int fixed = (fixed_half == 9 && rt > 4 && rt <= 8) ? 6 : fixed_half;
  return 2 * rt + fixed;
}

__host__ __device__ inline int svdd_choose_spt(int n, int L, int ncu, int fixed_half) {   // fixed_half: the fixed part in HALF row tiles
  const int smax = SVDD_TILE_ROWS / L;
  int best = smax;
  long long best_cost = -1;
  for (int s = smax; s >= 1; --s) {                   // ties: the fuller tile (fewer workgroups)
    const long long tiles = (n + s - 1) / s;
    const long long rounds = (tiles + ncu - 1) / ncu;
    const long long cost = rounds * svdd_tile_cost((s * L + 15) / 16, fixed_half);
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = s; }
  }
  return best;
}

// Round 5: TWO tile sizes per launch. With one size the launch costs whole rounds of it — 1100 live candidates of L = 50 are 367
// tiles of three = 2 rounds x 13 units although the second round is 43 % full. A plan is k FULL rounds (k x CUs tiles) of s1
// sequences followed by the remainder tiled at s2: 1100 = 256 tiles of four (16 units) + 76 single-sequence tiles (7 units) = 23.
// The single size is the k = 0 plan, so a plan never costs more than svdd_choose_spt's choice under the same model. The workgroups
// are dispatched in grid order, the large tiles first. A row's result does not depend on the plan.
struct SvddTilePlan { int s1, n1, s2; };          // tiles 0 .. n1 - 1 take s1 sequences each, the tiles after them s2

__host__ __device__ inline SvddTilePlan svdd_plan_tiles(int n, int L, int ncu, int fixed_half) {
  const int smax = SVDD_TILE_ROWS / L;
  SvddTilePlan best = {smax, 0, svdd_choose_spt(n, L, ncu, fixed_half)};
  long long best_cost;
  {
    const long long tiles = (n + best.s2 - 1) / best.s2;
    best_cost = ((tiles + ncu - 1) / ncu) * svdd_tile_cost((best.s2 * L + 15) / 16, fixed_half);
  }
  for (int s1 = smax; s1 >= 1; --s1) {
    const long long c1 = svdd_tile_cost((s1 * L + 15) / 16, fixed_half);
    const long long per_round = (long long)s1 * ncu;
    for (long long k = 1; k * per_round <= n; ++k) {
      const long long r = n - k * per_round;
      if (r == 0) {
        if (k * c1 < best_cost) { best_cost = k * c1; best = {s1, (int)(k * ncu), s1}; }
        continue;
      }
      for (int s2 = smax; s2 >= 1; --s2) {
        const long long tiles2 = (r + s2 - 1) / s2;
        const long long cost = k * c1 + ((tiles2 + ncu - 1) / ncu) * svdd_tile_cost((s2 * L + 15) / 16, fixed_half);
        if (cost < best_cost) { best_cost = cost; best = {s1, (int)(k * ncu), s2}; }
      }
    }
  }
  return best;
}

__host__ __device__ inline int svdd_plan_num_tiles(const SvddTilePlan& p, int n) {
  const int rest = n - p.n1 * p.s1;
  return p.n1 + (rest > 0 ? (rest + p.s2 - 1) / p.s2 : 0);
}

// first sequence and sequences of tile t
__host__ __device__ inline void svdd_plan_tile(const SvddTilePlan& p, int t, int& seq0, int& ns) {
  if (t < p.n1) { seq0 = t * p.s1; ns = p.s1; }
  else { seq0 = p.n1 * p.s1 + (t - p.n1) * p.s2; ns = p.s2; }
}
