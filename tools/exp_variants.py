"""Timing-only ablation experiments on the kernels: every variant deletes one component of a kernel (its results are
wrong by construction) so that the component's cost shows up as a time difference.

    python tools/exp_variants.py build <set>     here (no GPU needed): patched COPIES of svdd_amd/csrc are compiled into
                                                 build/exp/<set>/<variant>/libsvdd_hip.so (git-ignored, travels with gpurun)
    python tools/exp_variants.py run <set>       on the GPU box: the set's microbenchmark once per variant (SVDD_HIP_LIB)
    python tools/exp_variants.py check all       one line per set: VALID / PARTLY STALE / STALE against the current kernels
    python tools/exp_variants.py check <set>     which variants' patches still match the sources (older experiments go stale when
                                                 a kernel is restructured; their numbers stay in profiles/)

The tracked sources are never edited in place."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "svdd_amd", "csrc")

GRU_LP = {
    "file": "svdd_lp_gru_tail.hip",
    "bench": ["python", "tools/gru_lp_microbench.py", "f16x3", "2560", "2048", "1280"],
    "variants": {
        "baseline": [],
        "nogates": [("const float r = sigmoid_fast(acc_r[rho] * inv + b_r);", "const float r = acc_r[rho] * inv + b_r;"),
                    ("const float z = sigmoid_fast(acc_z[rho] * inv + b_z);", "const float z = acc_z[rho] * inv + b_z;"),
                    ("const float nn = tanh_fast(acc_nx[rho] * inv + b_nx + r * (acc_nh[rho] * inv + b_nh));",
                     "const float nn = acc_nx[rho] * inv + b_nx + r * (acc_nh[rho] * inv + b_nh);")],
        "nostore": [("if (seq0 + srow < n) a.out[", "if (seq0 + srow < n && hn[rho] == 12345.0f) a.out[")],
        "noprod_mfma": [("      if (s + 1 < L) project(1, 1);", "      if (s + 1 < L && n == 12345) project(1, 1);"),
                        ("      if (s + 2 < L) project(0, 0);", "      if (s + 2 < L && n == 12345) project(0, 0);")],
        "noprod_load": [("      if (s + 3 < L) load_x(t0 + (s + 3) * dt, 1);", "      if (s + 3 < L && n == 12345) load_x(t0 + (s + 3) * dt, 1);"),
                        ("      if (s + 4 < L) load_x(t0 + (s + 4) * dt, 0);", "      if (s + 4 < L && n == 12345) load_x(t0 + (s + 4) * dt, 0);")],
        "norec_mfma": [("#pragma unroll\n    for (int c = 0; c < 2; ++c) {\n      acc_nh = Lp<T>::mfma(hh[c], wb[2][c][0], acc_nh);",
                        "#pragma unroll\n    for (int c = 0; c < 2 && n == 12345; ++c) {\n      acc_nh = Lp<T>::mfma(hh[c], wb[2][c][0], acc_nh);")],
    },
}
GRU_LP["variants"]["nogates_nostore"] = GRU_LP["variants"]["nogates"] + GRU_LP["variants"]["nostore"]
GRU_LP["variants"]["noprod"] = GRU_LP["variants"]["noprod_mfma"] + GRU_LP["variants"]["noprod_load"]
GRU_LP["variants"]["only_barrier_lds"] = (GRU_LP["variants"]["nogates"] + GRU_LP["variants"]["nostore"] + GRU_LP["variants"]["noprod"] +
                                          GRU_LP["variants"]["norec_mfma"])
TOWER_LP = {
    "file": "svdd_lp_tower.hip",
    "bench": ["python", "tools/tower_lp_microbench.py", "f16x3", "3.0"],
    "variants": {
        "baseline": [],
        "noB": [("      if (it + 1 < nit) {\n        const V8* src = wsrc + (size_t)(it + 1) * TILE_V8;",
                 "      if (it + 1 < nit && a.n == 12345) {\n        const V8* src = wsrc + (size_t)(it + 1) * TILE_V8;")],
        "noA": [("          V[0] = *reinterpret_cast<const V8*>(plane + o_);                                                    \\\n          if constexpr (NP == 3) V[1] = *reinterpret_cast<const V8*>(plane + TPLANE_B + o_);                  \\",
                 "          V[0] = bc[0]; (void)o_;                                                                            \\\n          if constexpr (NP == 3) V[1] = bc[1];                                                                \\")],
        "nocopy": [("      if (row < keep_lo || row >= keep_hi) outc[e] = par[e];", "      if ((row < keep_lo || row >= keep_hi) && a.n == 12345) outc[e] = par[e];")],
        "nozero": [("  for (int e = tid; e < 2 * TPLANE_B / 4; e += 512) reinterpret_cast<int*>(smem_b)[e] = 0;",
                    "  for (int e = tid; e < 2 * TPLANE_B / 4 && a.n == 12345; e += 512) reinterpret_cast<int*>(smem_b)[e] = 0;")],
        "nomfma": [("          acc[R] = Lp<T>::mfma(U[0], bc[0], acc[R]);                                                          \\\n          if constexpr (NP == 3) { acc[R] = Lp<T>::mfma(U[0], bc[1], acc[R]); acc[R] = Lp<T>::mfma(U[1], bc[0], acc[R]); } \\",
                    "          acc[R][0] += (float)U[0][0] * (float)bc[0][0]; if constexpr (NP == 3) acc[R][1] += (float)U[1][0] * (float)bc[1][0];   \\\n          \\")],
    },
}
TOWER_LP["variants"]["noA_noB"] = TOWER_LP["variants"]["noA"] + TOWER_LP["variants"]["noB"]
TOWER_LP["variants"]["noA_noB_nomfma"] = TOWER_LP["variants"]["noA"] + TOWER_LP["variants"]["noB"] + TOWER_LP["variants"]["nomfma"]
BB_LP = {
    "file": "svdd_lp_backbone.hip",
    "bench": ["python", "tools/backbone_lp_check.py", "256", "200", "--time-only", "f16x3"],
    "variants": {
        "baseline": [],
        # LayerNorm statistics deleted: the image is written from f directly (still split + stored)
        "nostats": [("    if (layer < nl) {\n      const float tb0 = vl[BB_C + c0], tb1 = vl[BB_C + c0 + 1];", "    if (layer < nl && a.n == 12345) {\n      const float tb0 = vl[BB_C + c0], tb1 = vl[BB_C + c0 + 1];")],
        "nomfma": [("        acc[R][0] = Lp<T>::mfma(U[0], bc[0], acc[R][0]);                                                     \\\n        acc[R][1] = Lp<T>::mfma(U[0], bc[NPARTS], acc[R][1]);                                                \\",
                    "        acc[R][0][0] += (float)U[0][0] * (float)bc[0][0];                                                    \\\n        acc[R][1][0] += (float)U[0][1] * (float)bc[NPARTS][0];                                               \\"),
                   ("          acc[R][0] = Lp<T>::mfma(U[0], bc[1], acc[R][0]);                                                   \\\n          acc[R][1] = Lp<T>::mfma(U[0], bc[NPARTS + 1], acc[R][1]);                                          \\\n          acc[R][0] = Lp<T>::mfma(U[1], bc[0], acc[R][0]);                                                   \\\n          acc[R][1] = Lp<T>::mfma(U[1], bc[NPARTS], acc[R][1]);                                              \\",
                    "          acc[R][0][1] += (float)U[1][0] * (float)bc[1][0];                                                  \\")],
        "noA": [("        V[0] = *reinterpret_cast<const V8*>(plane + o_);                                                     \\\n        if constexpr (NP == 3) V[1] = *reinterpret_cast<const V8*>(plane + PLANE_B + o_); }",
                 "        V[0] = bc[0]; (void)o_;                                                                              \\\n        if constexpr (NP == 3) V[1] = bc[1]; }")],
        "noB": [("      if (nxt < it_end) {\n        const V8* src = wsrc + (size_t)tile_of(nxt) * TILE_V8;", "      if (nxt < it_end && a.n == 12345) {\n        const V8* src = wsrc + (size_t)tile_of(nxt) * TILE_V8;")],
        "nofirst": [("            if (tk >= 0) v += Bs[(t * 5 + tk) * BB_C + col];", "            if (tk >= 12345) v += Bs[(t * 5 + tk) * BB_C + col];")],
    },
}
BB_LP["variants"]["nodpp"] = [("  v += row_ror<8>(v); v += row_ror<4>(v); v += row_ror<2>(v); v += row_ror<1>(v);\n  return v;", "  return v;")]
BB_LP["variants"]["nobar"] = [("          if (j == 0 && row < TW_ROWS) psum[cg * TW_ROWS + row] = sm;\n        }\n      __syncthreads();", "          if (j == 0 && row < TW_ROWS) psum[cg * TW_ROWS + row] = sm;\n        }"),
                              ("          if (j == 0 && row < TW_ROWS) psum2[cg * TW_ROWS + row] = sq;     // second buffer: pass-1 partials may still be read\n        }\n      __syncthreads();", "          if (j == 0 && row < TW_ROWS) psum2[cg * TW_ROWS + row] = sq;\n        }")]
BB_LP["variants"]["nopsumread"] = [("          const float mean = row < TW_ROWS ? ((psum[row] + psum[TW_ROWS + row]) + (psum[2 * TW_ROWS + row] + psum[3 * TW_ROWS + row])) * (1.0f / BB_C) : 0.0f;", "          const float mean = 0.01f * sa;"),
                                   ("            const float rs = rsqrtf(((psum2[row] + psum2[TW_ROWS + row]) + (psum2[2 * TW_ROWS + row] + psum2[3 * TW_ROWS + row])) *\n                                    (1.0f / BB_C) + 1e-5f);", "            const float rs = 0.9f * sa;")]
BB_LP["variants"]["nopsumwrite"] = [("          if (j == 0 && row < TW_ROWS) psum[cg * TW_ROWS + row] = sm;", "          if (j == 0 && row < TW_ROWS && sm == 12345.0f) psum[cg * TW_ROWS + row] = sm;"),
                                    ("          if (j == 0 && row < TW_ROWS) psum2[cg * TW_ROWS + row] = sq;", "          if (j == 0 && row < TW_ROWS && sq == 12345.0f) psum2[cg * TW_ROWS + row] = sq;")]
BB_LP["variants"]["nomfma_noA_noB"] = BB_LP["variants"]["nomfma"] + BB_LP["variants"]["noA"] + BB_LP["variants"]["noB"]
BB_LP["variants"]["nostats_nomfma_noA_noB"] = BB_LP["variants"]["nostats"] + BB_LP["variants"]["nomfma_noA_noB"]
K1 = {
    "file": "svdd_kernels.hip",
    "bench": ["python", "tools/k1_one.py", "0.5"],
    "variants": {
        "baseline": [],
        "nostore": [("      a.cand[o] = (uint8_t)c;\n", "      if (c == 77) a.cand[o] = (uint8_t)c;\n"),
                    ("      __builtin_nontemporal_store(oh, reinterpret_cast<f32x4_t*>(a.onehot) + o);\n    };",
                     "      if (c == 77) __builtin_nontemporal_store(oh, reinterpret_cast<f32x4_t*>(a.onehot) + o);\n    };")],
        "plainstore": [("      __builtin_nontemporal_store(oh, reinterpret_cast<f32x4_t*>(a.onehot) + o);\n    };",
                        "      reinterpret_cast<f32x4_t*>(a.onehot)[o] = oh;\n    };")],
        "nophilox": [("            philox_uniform5(a.seed, (a.row_offset + (uint64_t)bs) * (uint64_t)a.L + (uint64_t)ls, a.step, (uint32_t)m, u);",
                      "            for (int v = 0; v < V; ++v) u[v] = (float)((ls * 5u + (uint32_t)v + (uint32_t)m * 977u) & 1023u) * (1.0f / 1024.0f);")],
        "nolog": [("              const float g = 1e-10f - log_fast(u[v] + 1e-10f);\n              const float r = qv[v] * __builtin_amdgcn_rcpf(g);",
                   "              const float g = 1.0f + u[v];\n              const float r = qv[v] * g;")],
        "nocandbyte": [("      a.cand[o] = (uint8_t)c;\n", "      if (c == 77) a.cand[o] = (uint8_t)c;\n")],
    },
}
K1["variants"]["nodraw"] = K1["variants"]["nophilox"] + K1["variants"]["nolog"]
K1["variants"]["nodraw_nostore"] = K1["variants"]["nodraw"] + K1["variants"]["nostore"]
TOWER2 = {
    "file": "svdd_nets.hip",
    "bench": ["python", "tools/tower_ab.py", "1", "3"],
    "variants": {
        "baseline": [],
        "nocopy": [("      if (row < keep_lo || row >= keep_hi)\n        *reinterpret_cast<float4*>(outc + (size_t)row * TW_C + 4 * (e & 15)) = *reinterpret_cast<const float4*>(par + (size_t)row * TW_C + 4 * (e & 15));",
                    "      if ((row < keep_lo || row >= keep_hi) && a.n == 12345)\n        *reinterpret_cast<float4*>(outc + (size_t)row * TW_C + 4 * (e & 15)) = *reinterpret_cast<const float4*>(par + (size_t)row * TW_C + 4 * (e & 15));")],
        "noB": [("      if (it + 1 < nit) {\n        const float* src = c.wsrc + (size_t)(it + 1) * TW_C * CH;",
                 "      if (it + 1 < nit && c.L == 12345) {\n        const float* src = c.wsrc + (size_t)(it + 1) * TW_C * CH;")],
        "nomfma": [("              acc[r][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s4], b[ct][4 * q + s4], acc[r][ct], 0, 0, 0);",
                    "              acc[r][ct][s4] += av[s4] * b[ct][4 * q + s4];")],
        "nozero": [("  for (int e = tid; e < (TW_ROWS + 2) * TW_AP; e += 512) smem[e] = 0.0f;      // image incl. the zero rows",
                    "  for (int e = tid; e < (TW_ROWS + 2) * TW_AP && a.n == 12345; e += 512) smem[e] = 0.0f;")],
        "nowriteback": [("      if (row >= keep_lo && row < keep_hi)\n        *reinterpret_cast<float4*>(outc + (size_t)row * TW_C + 4 * q) = *reinterpret_cast<const float4*>(act + (row - w0) * TW_AP + 4 * q);",
                         "      if (row >= keep_lo && row < keep_hi && a.n == 12345)\n        *reinterpret_cast<float4*>(outc + (size_t)row * TW_C + 4 * q) = *reinterpret_cast<const float4*>(act + (row - w0) * TW_AP + 4 * q);")],
    },
}
TOWER2["variants"]["nomfma0"] = [("              acc[r][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s4], b[ct][4 * q + s4], acc[r][ct], 0, 0, 0);",
                                  "              if (c.L == 12345) acc[r][ct][s4] += av[s4] * b[ct][4 * q + s4];")]
TOWER2["variants"]["noA"] = [("          const float4* ap = reinterpret_cast<const float4*>(actb + o);\n          af[r][0] = ap[0]; af[r][1] = ap[1];",
                              "          af[r][0] = bn[0]; af[r][1] = bn[1]; (void)o;")]
TOWER2["variants"]["noepi"] = [("          c.act[o] = row < c.tile_rows ? fmaxf(v, 0.0f) : 0.0f;", "          if (v == 12345.0f) c.act[o] = row < c.tile_rows ? fmaxf(v, 0.0f) : 0.0f;")]
TOWER2["variants"]["nolayers"] = [("  tower2_dispatch<SPT1 || WIN, WIN, CT>(c, nlive);", "  if (a.n == 12345) tower2_dispatch<SPT1 || WIN, WIN, CT>(c, nlive);")]
TOWER2["variants"]["nomfma0_noA_noB"] = TOWER2["variants"]["nomfma0"] + TOWER2["variants"]["noA"] + TOWER2["variants"]["noB"]
TOWER2["variants"]["nomfma0_noA_noB_noepi"] = TOWER2["variants"]["nomfma0_noA_noB"] + TOWER2["variants"]["noepi"]
TOWER2["variants"]["nocopy_nowriteback"] = TOWER2["variants"]["nocopy"] + TOWER2["variants"]["nowriteback"]
TOWER2["variants"]["nomfma_noB"] = TOWER2["variants"]["nomfma"] + TOWER2["variants"]["noB"]
GRU_PC = {
    "file": "svdd_nets.hip",
    "bench": ["python", "tools/gru_one.py", "2048", "2560"],
    "variants": {
        "baseline": [],
        "nogates": [("      const float r = sigmoid_fast(acc_r[rho]);\n      const float z = sigmoid_fast(acc_z[rho]);\n      const float nn = tanh_fast(acc_nx[rho] + r * acc_nh[rho]);\n      const float hn = (1.0f - z) * nn + z * hprev[rho];\n      hprev[rho] = hn;\n      const int srow = 4 * g + rho;\n      hbuf[cur ^ 1][srow][u] = hn;\n      if (seq0 + srow < n) out[",
                     "      const float r = acc_r[rho];\n      const float z = acc_z[rho];\n      const float nn = acc_nx[rho] + r * acc_nh[rho];\n      const float hn = (1.0f - z) * nn + z * hprev[rho];\n      hprev[rho] = hn;\n      const int srow = 4 * g + rho;\n      hbuf[cur ^ 1][srow][u] = hn;\n      if (seq0 + srow < n) out[")],
        "nostore": [("      hbuf[cur ^ 1][srow][u] = hn;\n      if (seq0 + srow < n) out[(((size_t)dir * n_alloc + seq0 + srow) * L + t) * H + u] = hn;\n    }\n    __syncthreads();\n  }\n}\n\n// ------------------------------------------------------------------ fused conv epilogue",
                     "      hbuf[cur ^ 1][srow][u] = hn;\n      if (seq0 + srow < n && hn == 12345.0f) out[(((size_t)dir * n_alloc + seq0 + srow) * L + t) * H + u] = hn;\n    }\n    __syncthreads();\n  }\n}\n\n// ------------------------------------------------------------------ fused conv epilogue")],
        "noprod_mfma": [("      if (s + 1 < L) project(1, 1);", "      if (s + 1 < L && n == 12345) project(1, 1);"),
                        ("      if (s + 2 < L) project(0, 0);", "      if (s + 2 < L && n == 12345) project(0, 0);")],
        "noprod_load": [("      load_x(t0 + min(s + 3, L - 1) * dt, 1);", "      if (n == 12345) load_x(t0 + min(s + 3, L - 1) * dt, 1);"),
                        ("      load_x(t0 + min(s + 4, L - 1) * dt, 0);", "      if (n == 12345) load_x(t0 + min(s + 4, L - 1) * dt, 0);")],
        "norec_mfma": [("    for (int s = 0; s < 16; ++s) {\n      acc_nh = __builtin_amdgcn_mfma_f32_16x16x4f32(ha[s], wr[2][s], acc_nh, 0, 0, 0);",
                        "    for (int s = 0; s < 16 && n == 12345; ++s) {\n      acc_nh = __builtin_amdgcn_mfma_f32_16x16x4f32(ha[s], wr[2][s], acc_nh, 0, 0, 0);")],
    },
}
GRU_PC["variants"]["nogates_nostore"] = GRU_PC["variants"]["nogates"] + GRU_PC["variants"]["nostore"]
GRU_PC["variants"]["noprod"] = GRU_PC["variants"]["noprod_mfma"] + GRU_PC["variants"]["noprod_load"]

# the transposed-accumulator split-precision backbone (round 3): kernel time by HIP events (not wall time)
BB_LPT = {
    "file": "svdd_lp_backbone.hip",
    "bench": ["python", "tools/backbone_lp_check.py", "256", "200", "--time-only", "f16x3", "--events"],
    "variants": {
        "baseline": [],
        "nostats": [("    if (layer < nl) {\n      f32x4 tb0, tb1;\n      chan8(vl + BB_C, tb0, tb1, 1.0f);", "    if (layer < nl && a.n == 12345) {\n      f32x4 tb0, tb1;\n      chan8(vl + BB_C, tb0, tb1, 1.0f);")],
        "nostatpass": [("#pragma unroll\n      for (int r = 0; r < NR; ++r) {\n        const int p = 16 * (rh + 2 * r) + j;\n        const float K = rmean[p < TW_ROWS ? p : TW_ROWS - 1];",
                        "      for (int r = 0; r < NR && a.n == 12345; ++r) {\n        const int p = 16 * (rh + 2 * r) + j;\n        const float K = rmean[p < TW_ROWS ? p : TW_ROWS - 1];")],
        "noxlane": [("        s1 += __shfl_xor(s1, 16, WAVE_SZ); s2 += __shfl_xor(s2, 16, WAVE_SZ);\n        s1 += __shfl_xor(s1, 32, WAVE_SZ); s2 += __shfl_xor(s2, 32, WAVE_SZ);\n", "")],
        "nomfma": [("        acc[R][0] = Lp<T>::mfma(bc[0], U[0], acc[R][0]);                                                     \\\n        acc[R][1] = Lp<T>::mfma(bc[NPARTS], U[0], acc[R][1]);                                                \\",
                    "        acc[R][0][0] += (float)U[0][0] * (float)bc[0][0];                                                    \\\n        acc[R][1][0] += (float)U[0][1] * (float)bc[NPARTS][0];                                               \\"),
                   ("          acc[R][0] = Lp<T>::mfma(bc[1], U[0], acc[R][0]);                                                   \\\n          acc[R][1] = Lp<T>::mfma(bc[NPARTS + 1], U[0], acc[R][1]);                                          \\\n          acc[R][0] = Lp<T>::mfma(bc[0], U[1], acc[R][0]);                                                   \\\n          acc[R][1] = Lp<T>::mfma(bc[NPARTS], U[1], acc[R][1]);                                              \\",
                    "          acc[R][0][1] += (float)U[1][0] * (float)bc[1][0];                                                  \\")],
        "noX": [("        V[0] = *reinterpret_cast<const V8*>(plane + o_);                                                     \\\n        if constexpr (NP == 3) V[1] = *reinterpret_cast<const V8*>(plane + PLANE_B + o_); }\n#define LPT_WAIT",
                 "        V[0] = bc[0]; (void)o_;                                                                              \\\n        if constexpr (NP == 3) V[1] = bc[1]; }\n#define LPT_WAIT")],
        "noW": [("      if (nxt < it_end) {\n        const V8* src = wsrc + (size_t)tile_of(nxt) * TILE_V8;\n#pragma unroll\n        for (int q = 0; q < 2 * NPARTS; ++q) bn[q] = src[q];\n      }\n      const int delta = (((en >> 15) & 15) - 4) * dil;\n      const int coff = ((en >> 13) & 3) * 64;\n      const int dbytes = delta * LPSB + coff;\n      const int live = en >> rh;                          // bit 2 r = owned tile r\n#define LPT_XLOAD",
                 "      if (nxt < it_end && a.n == 12345) {\n        const V8* src = wsrc + (size_t)tile_of(nxt) * TILE_V8;\n#pragma unroll\n        for (int q = 0; q < 2 * NPARTS; ++q) bn[q] = src[q];\n      }\n      const int delta = (((en >> 15) & 15) - 4) * dil;\n      const int coff = ((en >> 13) & 3) * 64;\n      const int dbytes = delta * LPSB + coff;\n      const int live = en >> rh;                          // bit 2 r = owned tile r\n#define LPT_XLOAD")],
        "nofirst": [("          va += tk >= 0 ? ta : z;\n          vb += tk >= 0 ? tb : z;", "          va += tk >= 12345 ? ta : z;\n          vb += tk >= 12345 ? tb : z;")],
        "noepi": [("        f[r][0] = __builtin_elementwise_max(acc[r][0] * inv + bl0, z4) + f[r][0];\n        f[r][1] = __builtin_elementwise_max(acc[r][1] * inv + bl1, z4) + f[r][1];",
                   "        f[r][0] = acc[r][0] + f[r][0];\n        f[r][1] = acc[r][1] + f[r][1];")],
    },
}
# scheduling experiments (correct results)
BB_LPT["variants"]["x_nofence"] = [("      __builtin_amdgcn_sched_barrier(0);                                                                     \\\n      LPT_WAIT(NOUT)                                                                                         \\\n".replace("\\\\", "\\"), "      \\\n".replace("\\\\", "\\")),
                                   ("      }                                                                                                      \\\n      __builtin_amdgcn_sched_barrier(0);\n".replace("\\\\", "\\"), "      }\n")]
BB_LPT["variants"]["nomfma_noX_noW"] = BB_LPT["variants"]["nomfma"] + BB_LPT["variants"]["noX"] + BB_LPT["variants"]["noW"]
BB_LPT["variants"]["nostats_nomfma_noX_noW"] = BB_LPT["variants"]["nostats"] + BB_LPT["variants"]["nomfma_noX_noW"]
BB_LPT["variants"]["noX_noW"] = BB_LPT["variants"]["noX"] + BB_LPT["variants"]["noW"]
# cycle counters at the phase boundaries of backbone_lp_t_kernel (correct results; read back by tools/lpt_phase_timing.py):
# per wave [0] LayerNorm finalize + image write incl. the "image complete" barrier, [1] the (tap, chunk) loop, [2] the wait at the
# first LayerNorm barrier (where the row groups meet again), [3] epilogue + statistics + first layer + last conv, [8 + layer] the loop of each layer
_TICK = "{ const unsigned long long t_ = __builtin_readcyclecounter(); tacc[%d] += t_ - tprev; tprev = t_; }"
BB_LPT_TIMING = {
    "file": "svdd_lp_backbone.hip",
    "bench": ["python", "tools/lpt_phase_timing.py"],
    "variants": {"timing": [
        ("typedef float f32x8 __attribute__((ext_vector_type(8)));",
         "typedef float f32x8 __attribute__((ext_vector_type(8)));\n__device__ unsigned long long g_lpt_dbg[256 * 8 * 32];"),
        ("  for (int layer = 0; layer <= nl; ++layer) {             // layer == nl: the first 1x1 conv of final_conv\n    const float* vl = a.vec + (size_t)(layer + 1) * 4 * BB_C;\n    const float sa = a.lscale[2 * layer], inv = a.lscale[2 * layer + 1];\n    // The phase code below",
         "  unsigned long long tacc[4] = {0, 0, 0, 0}, tprev = __builtin_readcyclecounter();\n  for (int layer = 0; layer <= nl; ++layer) {             // layer == nl: the first 1x1 conv of final_conv\n    " + _TICK % 3 + "\n    const float* vl = a.vec + (size_t)(layer + 1) * 4 * BB_C;\n    const float sa = a.lscale[2 * layer], inv = a.lscale[2 * layer + 1];\n    // The phase code below"),
        ("    __syncthreads();                                      // the image is complete\n    while (it < layer_end) {                              // one live tap",
         "    __syncthreads();                                      // the image is complete\n    " + _TICK % 0 + "\n    const unsigned long long tl0 = tprev;\n    while (it < layer_end) {                              // one live tap"),
        ("      it = nxt;\n      en = __builtin_amdgcn_readfirstlane(en_next_v);\n    }\n    // No barrier here between conv layers",
         "      it = nxt;\n      en = __builtin_amdgcn_readfirstlane(en_next_v);\n    }\n    " + _TICK % 1 + "\n    if (lane == 0 && blockIdx.x < 256) g_lpt_dbg[(blockIdx.x * 8 + w) * 32 + 8 + layer] = tprev - tl0;\n    // No barrier here between conv layers"),
        ("not seven (VGPR pressure)\n      }\n      __syncthreads();\n      if (tid < TW_ROWS) {",
         "not seven (VGPR pressure)\n      }\n      " + _TICK % 3 + "\n      __syncthreads();\n      " + _TICK % 2 + "\n      if (tid < TW_ROWS) {"),
        ("  __syncthreads();\n  // ---- last 1x1 conv 128 -> 5 in fp32\n  for (int e = tid; e < L * 5; e += NTH) {",
         "  __syncthreads();\n  " + _TICK % 3 + "\n  if (lane == 0 && blockIdx.x < 256) for (int k = 0; k < 4; ++k) g_lpt_dbg[(blockIdx.x * 8 + w) * 32 + k] = tacc[k];\n  // ---- last 1x1 conv 128 -> 5 in fp32\n  for (int e = tid; e < L * 5; e += NTH) {"),
        ("}  // namespace\n\nextern \"C\" int svdd_backbone_cnn_lp(",
         "}  // namespace\n\nextern \"C\" int svdd_internal_lpt_dbg(void* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_lpt_dbg), sizeof(g_lpt_dbg)); }\n\nextern \"C\" int svdd_backbone_cnn_lp("),
    ]},
}
_T = BB_LPT_TIMING["variants"]
_T["timing_noX"] = _T["timing"] + [("      { V[0] = *reinterpret_cast<const V8*>(plane + xa[R] + 64 * (C));", "      if (a.n == 12345) { V[0] = *reinterpret_cast<const V8*>(plane + xa[R] + 64 * (C));")]
_T["timing_noW"] = _T["timing"] + [("      if (COND) { const V8* src_ = wsrc + (size_t)(TILE) * TILE_V8;", "      if ((COND) && a.n == 12345) { const V8* src_ = wsrc + (size_t)(TILE) * TILE_V8;")]
_T["timing_noX_noW"] = _T["timing_noX"] + _T["timing_noW"][-1:]
_T["timing_solo0"] = _T["timing"] + [("      if constexpr (NR == 6) LPT_TAP(6)", "      if constexpr (NR == 6) { if (a.n == 12345) LPT_TAP(6) }")]
_T["timing_solo1"] = _T["timing"] + [("        if (rg == 0) {                          // 7 row tiles", "        if (rg == 0 && a.n == 12345) {                          // 7 row tiles"),
                                      ("        } else if (rg == 1) {                          // 6 row tiles", "        } else if (rg == 1) { if (a.n != 12345) {                         // 6 row tiles"),
                                      ("          LPT_STEP(5, ub, wB, )\n        }\n      }\n      if constexpr (RG == 3) {", "          LPT_STEP(5, ub, wB, )\n        } }\n      }\n      if constexpr (RG == 3) {")]
# every step's MFMAs unconditional (dead tiles read the zero rows: same results, more MFMAs in the dilation-64 layers)
_NOBR = [("        if (lv_ & (1 << (RG * (R)))) acc[R][0] = Lp<T>::mfma(W[0], U[0], acc[R][0]);", "        acc[R][0] = Lp<T>::mfma(W[0], U[0], acc[R][0]);"),
         ("        asm volatile(\"\" : \"+s\"(lv_));                                                                        \\\n        if (lv_ & (1 << (RG * (R)))) {", "        {")]
_T["timing_nobranch"] = _T["timing"] + _NOBR
_T["timing_nobranch_solo0"] = _T["timing_solo0"] + _NOBR
_T["timing_nowait"] = _T["timing"] + [("      __builtin_amdgcn_s_waitcnt(0xC07F);                   /* lgkmcnt(0): this step's fragments */          \\\n", "")]
_T["timing_nobranch_noX_solo0"] = _T["timing_solo0"] + _NOBR + _T["timing_noX"][-1:]
# finer: [4] cycles in the steps between two weight prefetches, [5] in the prefetch (address + 4 loads + counted wait), [6] per tap before the first fragment request
_FINE = [("      if (COND) { const V8* src_ = wsrc + (size_t)(TILE) * TILE_V8;", "      { const unsigned long long t_ = __builtin_readcyclecounter(); tf[0] += t_ - tfp; tfp = t_; }                            \\\n      if (COND) { const V8* src_ = wsrc + (size_t)(TILE) * TILE_V8;"),
         ("      __builtin_amdgcn_s_waitcnt(0x0F70 | (2 * NPARTS));\n", "      __builtin_amdgcn_s_waitcnt(0x0F70 | (2 * NPARTS));                                                     \\\n      { const unsigned long long t_ = __builtin_readcyclecounter(); tf[1] += t_ - tfp; tfp = t_; }\n"),
         ("      const int nxt = en >> 19;\n      const int en_next_v = sched[nxt < it_end ? nxt : it];\n      const int delta = (((en >> 15) & 15) - 4) * dil;\n      const int live = en >> rg;",
          "      { const unsigned long long t_ = __builtin_readcyclecounter(); tf[0] += t_ - tfp; tfp = t_; }\n      const int nxt = en >> 19;\n      const int en_next_v = sched[nxt < it_end ? nxt : it];\n      const int delta = (((en >> 15) & 15) - 4) * dil;\n      const int live = en >> rg;"),
         ("      V8 ua[2], ub[2];                                    // activation fragments", "      { const unsigned long long t_ = __builtin_readcyclecounter(); tf[2] += t_ - tfp; tfp = t_; }\n      V8 ua[2], ub[2];                                    // activation fragments"),
         ("  unsigned long long tacc[4] = {0, 0, 0, 0}, tprev = __builtin_readcyclecounter();", "  unsigned long long tacc[4] = {0, 0, 0, 0}, tprev = __builtin_readcyclecounter(), tf[3] = {0, 0, 0}, tfp = 0;"),
         ("    const unsigned long long tl0 = tprev;", "    const unsigned long long tl0 = tprev; tfp = tprev;"),
         ("for (int k = 0; k < 4; ++k) g_lpt_dbg[(blockIdx.x * 8 + w) * 32 + k] = tacc[k];", "for (int k = 0; k < 7; ++k) g_lpt_dbg[(blockIdx.x * 8 + w) * 32 + k] = k < 4 ? tacc[k] : tf[k - 4];")]
_T["fine"] = _T["timing"] + _FINE
_T["fine_solo0"] = _T["timing_solo0"] + _FINE
_T["fine_nobranch_solo0"] = _T["timing_solo0"] + _NOBR + _FINE
# time stamps of every step of one tap (SVDD_STAMP_LAYER / SVDD_STAMP_TAP at build time, workgroups 0 and 1): slot 64 * wave + n in g_lpt_dbg[8192 ...]
_STAMP = "if (dbg_on) { const unsigned long long t_ = __builtin_readcyclecounter(); if (lane == 0 && cnt < 62) g_lpt_dbg[8192 + (blockIdx.x * 8 + w) * 64 + cnt] = t_; ++cnt; }"
def _stamps(layer, tap):
    return [("#define LPT_STEP_ALL(R, U, W, LOADNEXT, EXTRA)                                                               \\\n      __builtin_amdgcn_sched_barrier(0);                                                                     \\\n      __builtin_amdgcn_s_waitcnt(0xC07F);                   /* lgkmcnt(0): this step's fragments */          \\\n",
             "#define LPT_STEP_ALL(R, U, W, LOADNEXT, EXTRA)                                                               \\\n      __builtin_amdgcn_sched_barrier(0);                                                                     \\\n      __builtin_amdgcn_s_waitcnt(0xC07F);                   /* lgkmcnt(0): this step's fragments */          \\\n      " + _STAMP + " \\\n"),
            ("#define LPT_STEP_LIVE(R, U, W, LOADNEXT, EXTRA)                                                              \\\n      __builtin_amdgcn_sched_barrier(0);                                                                     \\\n",
             "#define LPT_STEP_LIVE(R, U, W, LOADNEXT, EXTRA)                                                              \\\n      __builtin_amdgcn_sched_barrier(0);                                                                     \\\n      " + _STAMP + " \\\n"),
            ("      V8 ua[2], ub[2];                                    // activation fragments", "      const bool dbg_on = blockIdx.x < 2 && layer == %d && ((en >> 15) & 15) == %d;\n      int cnt = 0;\n      " % (layer, tap) + _STAMP + "\n      V8 ua[2], ub[2];                                    // activation fragments"),
            ("      it = nxt;\n      en = __builtin_amdgcn_readfirstlane(en_next_v);\n    }\n    __syncthreads();                                      // every wave is done reading the image\n    f32x4 bl0, bl1;",
             "      " + _STAMP + "\n      it = nxt;\n      en = __builtin_amdgcn_readfirstlane(en_next_v);\n    }\n    __syncthreads();                                      // every wave is done reading the image\n    f32x4 bl0, bl1;")]
_T["stamps_d1"] = [_T["timing"][0], _T["timing"][-1]] + _stamps(2, 4)
_T["stamps_d64"] = [_T["timing"][0], _T["timing"][-1]] + _stamps(17, 5)
_T["stamps_d64b"] = [_T["timing"][0], _T["timing"][-1]] + _stamps(17, 6)
# the same boundaries in the exact-fp32 backbone_kernel (svdd_nets.hip): [0] LayerNorm (two passes, four barriers) + image write +
# "image complete" barrier, [1] the (chunk, tap) loop, [2] the barrier after it, [3] epilogue, first layer, last conv
BB_F32_TIMING = {
    "file": "svdd_nets.hip",
    "first_only": True,      # backbone_grad_kernel (round 5) repeats backbone_kernel's loop text further down: patch the first occurrence only
    "bench": ["python", "tools/lpt_phase_timing.py", "f32"],
    "variants": {"timing": [
        ("struct BackboneArgs {", "__device__ unsigned long long g_lpt_dbg[256 * 8 * 32];\nstruct BackboneArgs {"),
        ("  for (int layer = 0; layer <= nl; ++layer) {             // layer == nl: the first 1x1 conv of final_conv\n    const float* vl = a.vec + (size_t)(layer + 1) * 4 * BB_C;\n    if (layer < nl) {\n      const float tb0 = vl[BB_C + col0], tb1 = vl[BB_C + col0 + 16];",
         "  unsigned long long tacc[4] = {0, 0, 0, 0}, tprev = __builtin_readcyclecounter();\n  for (int layer = 0; layer <= nl; ++layer) {             // layer == nl: the first 1x1 conv of final_conv\n    " + _TICK % 3 + "\n    const float* vl = a.vec + (size_t)(layer + 1) * 4 * BB_C;\n    if (layer < nl) {\n      const float tb0 = vl[BB_C + col0], tb1 = vl[BB_C + col0 + 16];"),
        ("    __syncthreads();                                      // the image is complete\n    // A fragments of row tiles 0 and 1 of an entry are requested during the LAST MFMA groups",
         "    __syncthreads();                                      // the image is complete\n    " + _TICK % 0 + "\n    const unsigned long long tl0 = tprev;\n    // A fragments of row tiles 0 and 1 of an entry are requested during the LAST MFMA groups"),
        ("#undef B2_ALOAD\n    __syncthreads();                                      // every wave is done reading the image\n",
         "#undef B2_ALOAD\n    " + _TICK % 1 + "\n    if (lane == 0 && blockIdx.x < 256) g_lpt_dbg[(blockIdx.x * 8 + w) * 32 + 8 + layer] = tprev - tl0;\n    __syncthreads();                                      // every wave is done reading the image\n    " + _TICK % 2 + "\n"),
        ("  __syncthreads();\n  // ---- last 1x1 conv 128 -> 5: one (row, class) dot product per thread iteration\n  for (int e = tid; e < tile_rows * 5; e += 512) {",
         "  __syncthreads();\n  " + _TICK % 3 + "\n  if (lane == 0 && blockIdx.x < 256) for (int k = 0; k < 4; ++k) g_lpt_dbg[(blockIdx.x * 8 + w) * 32 + k] = tacc[k];\n  // ---- last 1x1 conv 128 -> 5: one (row, class) dot product per thread iteration\n  for (int e = tid; e < tile_rows * 5; e += 512) {"),
        ("extern \"C\" int svdd_backbone_cnn_f32(", "extern \"C\" int svdd_internal_lpt_dbg(void* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_lpt_dbg), sizeof(g_lpt_dbg)); }\n\nextern \"C\" int svdd_backbone_cnn_f32("),
    ]},
}
_SKIPLOOP = "while (it < layer_end) { const int nxt_ = en >> 19; en = __builtin_amdgcn_readfirstlane(sched[nxt_ < it_end ? nxt_ : it]); it = nxt_; }"
_F = BB_F32_TIMING["variants"]
_F["timing_solo0"] = _F["timing"] + [("    } else {\n      while (it < layer_end) {\n        B2_ENTRY(ua, ub, bA, bB)", "    } else {\n      if (a.n != 12345) { " + _SKIPLOOP + " }\n      while (it < layer_end) {\n        B2_ENTRY(ua, ub, bA, bB)")]
_F["timing_solo1"] = _F["timing"] + [("    if (rh == 0) {\n      while (it < layer_end) {\n        B2_ENTRY(ua, ub, bA, bB)", "    if (rh == 0) {\n      if (a.n != 12345) { " + _SKIPLOOP + " }\n      while (it < layer_end) {\n        B2_ENTRY(ua, ub, bA, bB)")]
_F["timing_nobranch"] = _F["timing"] + [("      if (live & (1 << (2 * (R)))) {                                                                         \\\n        _Pragma(\"unroll\") for (int q = 0; q < 2; ++q) {                                                      \\\n          acc[R][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(U[q].x, bf0[4 * q], acc[R][0], 0, 0, 0);          \\",
                                          "      {                                                                         \\\n        _Pragma(\"unroll\") for (int q = 0; q < 2; ++q) {                                                      \\\n          acc[R][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(U[q].x, bf0[4 * q], acc[R][0], 0, 0, 0);          \\")]
_FSTAMP = "if (dbg_on) { const unsigned long long t_ = __builtin_readcyclecounter(); if (lane == 0 && cnt < 62) g_lpt_dbg[8192 + (blockIdx.x * 8 + w) * 64 + cnt] = t_; ++cnt; }"
_F["stamps"] = [_F["timing"][0], _F["timing"][-1],
    ("#define B2_MM(R, U, NOUT)                                                                                    \\\n      __builtin_amdgcn_sched_barrier(0);                                                                     \\\n      B2_WAIT(NOUT)                                                                                          \\\n",
     "#define B2_MM(R, U, NOUT)                                                                                    \\\n      __builtin_amdgcn_sched_barrier(0);                                                                     \\\n      B2_WAIT(NOUT)                                                                                          \\\n      " + _FSTAMP + " \\\n"),
    ("    float4 ua[2], ub[2];\n    if (it < layer_end) {\n      B2_PARAMS(en, delta0, coff0, dbytes0)", "    const bool dbg_on = blockIdx.x < 2 && layer == 2;\n    int cnt = 0;\n    float4 ua[2], ub[2];\n    if (it < layer_end) {\n      B2_PARAMS(en, delta0, coff0, dbytes0)")]
_F["stamps_solo0"] = _F["stamps"] + _F["timing_solo0"][-1:]
_F["timing_solo0_noX"] = _F["timing_solo0"] + [("        V[0] = ap_[0]; V[1] = ap_[1]; }\n#define B2_WAIT", "        if (a.n == 12345) { V[0] = ap_[0]; V[1] = ap_[1]; } }\n#define B2_WAIT")]
_F["timing_solo0_nobranch"] = _F["timing_solo0"] + _F["timing_nobranch"][-1:]
_F["timing_solo0_noW"] = _F["timing_solo0"] + [("      if (nxt < it_end) {                                                                                    \\\n        const float* src = wsrc + (size_t)tile_of(nxt) * BB_C * CH;", "      if (nxt < it_end && a.n == 12345) {                                                                                    \\\n        const float* src = wsrc + (size_t)tile_of(nxt) * BB_C * CH;")]
_F["timing_noW"] = _F["timing"] + [("        const float* src = wsrc + (size_t)tile_of(nxt < it_end ? nxt : it) * BB_C * CH;                      \\\n", "        const float* src = wsrc + (size_t)(a.n == 12345 ? tile_of(nxt < it_end ? nxt : it) : 0) * BB_C * CH;                      \\\n")]
_F["timing_noX"] = _F["timing"] + [("        V[0] = make_float4(t0_[0], t0_[1], t0_[2], t0_[3]); V[1] = make_float4(t1_[0], t1_[1], t1_[2], t1_[3]); }", "        if (a.n == 12345) { V[0] = make_float4(t0_[0], t0_[1], t0_[2], t0_[3]); V[1] = make_float4(t1_[0], t1_[1], t1_[2], t1_[3]); } }")]
# cycle stamps in tower_lp_kernel (window kernel): per wave [0] prologue (parent rows copied, one-hot, planes zeroed), [1] MFMA loops,
# [2] the barrier after a layer's loop, [3] epilogues + "image complete" barriers, [4] output copy
_TT = "{ const unsigned long long t_ = __builtin_readcyclecounter(); tacc[%d] += t_ - tprev; tprev = t_; }"
TOWER_LP_TIMING = {
    "file": "svdd_lp_tower.hip",
    "bench": ["python", "tools/tower_lp_timing.py"],
    "variants": {"timing": [
        ("constexpr int TW_C = 64;", "__device__ unsigned long long g_tw_dbg[4096 * 8 * 8];\n__device__ unsigned long long g_tw_acc[8];\nconstexpr int TW_C = 64;"),
        ("  int it = 0;\n  f32x4 acc[NL > 0 ? NL : 1][2];\n", "  int it = 0;\n  f32x4 acc[NL > 0 ? NL : 1][2];\n  unsigned long long tacc[4] = {0, 0, 0, 0}, tprev = __builtin_readcyclecounter();\n"),
        ("    // every wave must be done reading the image before its owners overwrite it (the stem reads xs, not the image)\n    if (layer >= 0) __syncthreads();\n",
         "    " + _TT % 1 + "\n    if (layer >= 0) __syncthreads();\n    " + _TT % 2 + "\n"),
        ("    __syncthreads();                                             // the image is complete\n  }\n}",
         "    __syncthreads();                                             // the image is complete\n    " + _TT % 3 + "\n  }\n  if ((threadIdx.x & 63) == 0) for (int k = 1; k < 4; ++k) g_tw_dbg[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + k] = tacc[k];\n}"),
        ("  const int nlive = WIN ? (nt - c.rq + 3) >> 2 : (c.rq == 0 ? 4 : 3);    // owned live tiles rq + 4 r, r < nlive\n  __syncthreads();",
         "  const int nlive = WIN ? (nt - c.rq + 3) >> 2 : (c.rq == 0 ? 4 : 3);    // owned live tiles rq + 4 r, r < nlive\n  __syncthreads();\n  const unsigned long long tp1 = __builtin_readcyclecounter();\n  if (lane == 0) g_tw_dbg[(blockIdx.x * 8 + w) * 8 + 0] = tp1 - tp0;"),
        ("  const int tid = threadIdx.x, lane = tid & 63;\n  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);\n  const int L = a.L;\n\n  int cand = blockIdx.x;",
         "  const unsigned long long tp0 = __builtin_readcyclecounter();\n  const int tid = threadIdx.x, lane = tid & 63;\n  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);\n  const int L = a.L;\n\n  int cand = blockIdx.x;"),
        ("  // the last image IS the output: rows [hi 128 B | lo 128 B], 16 bytes per thread\n  if (WIN) {",
         "  const unsigned long long tp2 = __builtin_readcyclecounter();\n  // the last image IS the output: rows [hi 128 B | lo 128 B], 16 bytes per thread\n  if (WIN) {"),
        ("        outc[(row0 + row) * ROW16 + q] = *reinterpret_cast<const uint4*>(plane + (q >> 3) * TPLANE_B + row * TLSB + 16 * (q & 7));\n    }\n  }\n}",
         "        outc[(row0 + row) * ROW16 + q] = *reinterpret_cast<const uint4*>(plane + (q >> 3) * TPLANE_B + row * TLSB + 16 * (q & 7));\n    }\n  }\n  if (lane == 0) { const unsigned long long te = __builtin_readcyclecounter(); g_tw_dbg[(blockIdx.x * 8 + w) * 8 + 4] = te - tp2; g_tw_dbg[(blockIdx.x * 8 + w) * 8 + 5] = te - tp0; g_tw_dbg[(blockIdx.x * 8 + w) * 8 + 6] = 1; }\n}"),
        ("}  // namespace\n", "}  // namespace\n\nextern \"C\" int svdd_internal_tw_dbg(void* dst, int reset) { if (reset) return (int)hipMemset((void*)((char*)0 + 0), 0, 0) * 0 + (int)hipMemcpyToSymbol(HIP_SYMBOL(g_tw_dbg), dst, sizeof(g_tw_dbg)); return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_tw_dbg), sizeof(g_tw_dbg)); }\n"),
    ]},
}
# round 6: is the per-tile FIXED cost of the fp32 backbone on short sequences (4.5 row tiles' worth, csrc/svdd_spt.h) the latency of the
# weight stream (one tile requested one entry ahead; an entry of a 4-row-tile workgroup is ~1300 cycles of MFMA work)? "noW": every entry
# reads weight tile 0 (cache-hot); "base": the tracked kernel. Timing only (noW computes garbage). Bench: the calibration launches.
BB_F32_SMALL = {
    "file": "svdd_nets.hip",
    "bench": ["python", "tools/backbone_spt_calib.py", "f32"],
    "variants": {"base": [], "noW": _F["timing_noW"][-1:]},
}
# round 6: do the headline-config parity tests have teeth? The fp32 backbone's LayerNorm epsilon moved from 1e-5 by
# 20 % / 2 x / 10 x (logits then differ from the reference's by ~3e-5 / 1e-4 / 1e-3): tools/perturbation_check.sh runs the C2 / C3 /
# TDS / un-guided reference-run tests against each build and records which turn red (profiles/r06_perturbation_check.txt).
_EPS = "(1.0f / BB_C) + 1e-5f);\n      __syncthreads();\n      const float gm0 = vl[2 * BB_C + col0]"
PERTURB = {
    "file": "svdd_nets.hip",
    "first_only": True,
    "bench": ["true"],
    "variants": {name: [(_EPS, _EPS.replace("1e-5f", val))] for name, val in (("eps1.2", "1.2e-5f"), ("eps2", "2e-5f"), ("eps10", "1e-4f"))},
}
TAIL = {
    "file": "svdd_nets.hip",
    "bench": ["python", "tools/tail_microbench.py"],
    "variants": {
        "baseline": [],
        "nomfma": [("      for (int ct = 0; ct < 8; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[sidx], wb[16 * ct + sidx], acc[ct], 0, 0, 0);",
                    "      for (int ct = 0; ct < 8; ++ct) acc[ct][sidx & 3] += v[sidx] * wb[16 * ct + sidx];")],
        "noln": [("    sm += __shfl_xor(sm, 16, 64); sm += __shfl_xor(sm, 32, 64);\n    const float mean = sm * (1.0f / 64.0f);", "    const float mean = sm * (1.0f / 64.0f);"),
                 ("    sq += __shfl_xor(sq, 16, 64); sq += __shfl_xor(sq, 32, 64);\n    const float rstd = rsqrtf(sq * (1.0f / 64.0f) + 1e-5f);", "    const float rstd = sq * (1.0f / 64.0f) + 1e-5f;")],
        "noload": [("    load_rows(min(tile + 1, ntiles - 1));                 // fly under", "    if (n == 12345) load_rows(min(tile + 1, ntiles - 1));                 // fly under")],
        "noepi": [("          const float z = fmaxf(acc[ct][rho], 0.0f);\n#pragma unroll\n          for (int t = 0; t < T; ++t) part[t] += z * (HOIST ? wl[ct * T + t] : cv[8 + ct * T + t]);",
                   "          part[0] += acc[ct][rho];")],      # every accumulator is still read (the MFMAs stay): no max, no multiply
        "lb1": [("template <int T>\n__global__ __launch_bounds__(256, 2) void value_tail_kernel(", "template <int T>\n__global__ __launch_bounds__(256, 1) void value_tail_kernel(")],
    },
}
for _nm, _cond in (("skew_hi", "((blockIdx.x >> 8) & 1)"), ("skew_lo", "(blockIdx.x & 1)"), ("skew_wave", "((threadIdx.x >> 6) & 1)")):
    TAIL["variants"][_nm] = [("  load_rows(0);\n  for (int tile = 0; tile < ntiles; ++tile) {\n    float v[16];",
                              "  load_rows(0);\n  if (" + _cond + ") __builtin_amdgcn_s_sleep(56);\n  for (int tile = 0; tile < ntiles; ++tile) {\n    float v[16];")]
TAIL["variants"]["nomfma_noln_noepi"] = TAIL["variants"]["nomfma"] + TAIL["variants"]["noln"] + TAIL["variants"]["noepi"]
SETS = {"tail": TAIL, "perturb": PERTURB, "bb_f32_small": BB_F32_SMALL, "tower_lp_timing": TOWER_LP_TIMING, "bb_f32_timing": BB_F32_TIMING, "bb_lpt_timing": BB_LPT_TIMING, "bb_lpt": BB_LPT, "gru_pc": GRU_PC, "gru_lp": GRU_LP, "tower_lp": TOWER_LP, "bb_lp": BB_LP, "k1": K1, "tower2": TOWER2}


def build_variant(setname, name, spec):
    work = os.path.join(ROOT, "build", "exp", setname, name)
    shutil.rmtree(work, ignore_errors=True)
    os.makedirs(work)
    for f in os.listdir(CSRC):
        if f.endswith(".hip") or f.endswith(".h") or f == "Makefile":
            shutil.copy(os.path.join(CSRC, f), work)
    for f in os.listdir(CSRC):                       # objects of the untouched sources are reused
        if f.endswith(".o") and f[:-2] + ".hip" != spec["file"]:
            shutil.copy(os.path.join(CSRC, f), work)
            os.utime(os.path.join(work, f))
    p = os.path.join(work, spec["file"])
    s = open(p).read()
    for old, new in spec["variants"][name]:
        if old not in s:                                  # the variant was written against an earlier revision of the kernel
            shutil.rmtree(work, ignore_errors=True)
            return name, False, "STALE: its patch no longer matches the source (the experiment's numbers are in profiles/; `check` lists these)"
        s = s.replace(old, new, 1) if spec.get("first_only") else s.replace(old, new)
    open(p, "w").write(s)
    r = subprocess.run(["make", "-C", work, "INC=" + os.path.join(ROOT, "include")], capture_output=True, text=True)
    ok = os.path.exists(os.path.join(work, "libsvdd_hip.so"))
    for f in os.listdir(work):                       # keep only the library
        if f != "libsvdd_hip.so":
            os.remove(os.path.join(work, f))
    return name, ok, r.stderr[-400:] if not ok else ""


def _applies(src, patches):
    t = src
    for old, new in patches:
        if old not in t:
            return False
        t = t.replace(old, new)
    return True


def main():
    cmd, setname = sys.argv[1], sys.argv[2]
    if cmd == "check" and setname == "all":              # every set: how many of its variants still apply to today's kernels
        for sn, sp in SETS.items():
            src = open(os.path.join(CSRC, sp["file"])).read()
            ok = [n for n, patches in sp["variants"].items() if _applies(src, patches)]
            stale = [n for n in sp["variants"] if n not in ok]
            state = "VALID (every variant applies)" if not stale else ("STALE (none applies: numbers in profiles/ only)" if not ok else
                                                                       f"PARTLY STALE ({len(ok)} of {len(sp['variants'])} apply)")
            print(f"{sn:20s} {sp['file']:24s} {state}" + (f"   stale: {', '.join(stale)}" if stale and ok else ""))
        return
    spec = SETS[setname]
    only = sys.argv[3:]
    names = [n for n in spec["variants"] if not only or n in only]
    if cmd == "check":                                   # which variants still apply to the current sources
        src = open(os.path.join(CSRC, spec["file"])).read()
        for n in names:
            t, ok = src, True
            for old, new in spec["variants"][n]:
                ok = ok and old in t
                t = t.replace(old, new)
            print(f"{n:32s} {'applies' if ok else 'STALE'}")
        return
    if cmd == "build":
        with ThreadPoolExecutor(4) as ex:
            for name, ok, err in ex.map(lambda n: build_variant(setname, n, spec), names):
                print(name, "ok" if ok else "FAILED " + err)
    else:
        for name in names:
            lib = os.path.join(ROOT, "build", "exp", setname, name, "libsvdd_hip.so")
            env = dict(os.environ, SVDD_HIP_LIB=lib)
            r = subprocess.run(spec["bench"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
            lines = [ln for ln in r.stdout.splitlines() if "amdgpu.ids" not in ln]
            print(f"{name:28s} " + " | ".join(lines) + (" ERR " + r.stderr[-200:] if r.returncode else ""))


if __name__ == "__main__":
    main()
