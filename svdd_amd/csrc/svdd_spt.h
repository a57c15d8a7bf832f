// svdd_spt.h — how many whole sequences a backbone workgroup takes when several fit its 208-row tile (L <= 104).
// A tile costs its row tiles of 16 plus a fixed part (the loop skeleton of the 756 (layer, chunk, tap) iterations, LayerNorm
// phases, barriers, the first-layer lookup): measured with 256 x s sequences of L = 50 / 33 (tools/backbone_spt_calib.py,
// one full round): fp32 865 / 1240 / 1582 / 1941 us at 4 / 7 / 10 / 13 row tiles = 120 us per row tile + 3 row tiles' worth
// fixed; f16x3 483 / 585 / 655 / 733 us = 28 us per row tile + 13 row tiles' worth fixed (`fixed`, passed by the caller).
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

__host__ __device__ inline int svdd_choose_spt(int n, int L, int ncu, int fixed) {
  const int smax = SVDD_TILE_ROWS / L;
  int best = smax;
  long long best_cost = -1;
  for (int s = smax; s >= 1; --s) {                   // ties: the fuller tile (fewer workgroups)
    const long long tiles = (n + s - 1) / s;
    const long long rounds = (tiles + ncu - 1) / ncu;
    const long long cost = rounds * ((s * L + 15) / 16 + fixed);
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = s; }
  }
  return best;
}
