#!/bin/bash
# timing-only experiments on gru_bidir_kernel (results wrong by construction)
# The tracked sources are never touched: the kernels are copied to a scratch directory, patched and built THERE, and the
# microbenchmark loads that build through SVDD_HIP_LIB (svdd_amd/_lib.py). The scratch directory is removed on any exit.
cd "$(dirname "$0")/.." || exit 1
ROOT=$PWD
WORK=$(mktemp -d /tmp/svdd_exp.XXXXXX)
trap 'rm -rf "$WORK"' EXIT
fresh() { cp svdd_amd/csrc/*.hip svdd_amd/csrc/Makefile "$WORK"/; }
build() { make -C "$WORK" -j3 INC="$ROOT/include" 2>&1 | grep -E " error"; }
run() { build; echo "$1: $(SVDD_HIP_LIB=$WORK/libsvdd_hip.so timeout 120 python tools/gru_microbench.py 16 2048 2560 2>&1 | grep 'mode=0' | tr '\n' ' ')"; }
edit() { python3 - "$WORK" "$@" <<'PY'
import sys
p=sys.argv[1]+'/svdd_nets.hip'
s=open(p).read()
a=s.index("template <bool BOTH>"); b=s.index("// ------------------------------------------------------------------ fused conv epilogue + LayerNorm ----")
k=s[a:b]
for e in sys.argv[2:]:
    if e=="nomfma":
        k=k.replace("__builtin_amdgcn_mfma_f32_16x16x4f32(ha[s], wr[80 + s], acc_nh, 0, 0, 0)","acc_nh + ha[s] * wr[80+s]")
        k=k.replace("__builtin_amdgcn_mfma_f32_16x16x4f32(ha[s], wr[16 + s], acc_r, 0, 0, 0)","acc_r + ha[s] * wr[16+s]")
        k=k.replace("__builtin_amdgcn_mfma_f32_16x16x4f32(ha[s], wr[48 + s], acc_z, 0, 0, 0)","acc_z + ha[s] * wr[48+s]")
        k=k.replace("__builtin_amdgcn_mfma_f32_16x16x4f32(xn[s], wr[s], acc_r, 0, 0, 0)","acc_r + xn[s] * wr[s]")
        k=k.replace("__builtin_amdgcn_mfma_f32_16x16x4f32(xn[s], wr[32 + s], acc_z, 0, 0, 0)","acc_z + xn[s] * wr[32+s]")
        k=k.replace("__builtin_amdgcn_mfma_f32_16x16x4f32(xn[s], wr[64 + s], acc_nx, 0, 0, 0)","acc_nx + xn[s] * wr[64+s]")
    if e=="nostore":
        k=k.replace("if (srow < ts && seq0 + srow < n) out[","if (srow < ts && seq0 + srow < n && hn == 12345.0f) out[")
    if e=="nogates":
        k=k.replace("const float r = sigmoid_fast(acc_r[rho]);","const float r = acc_r[rho];").replace("const float z = sigmoid_fast(acc_z[rho]);","const float z = acc_z[rho];").replace("const float nn = tanh_fast(acc_nx[rho] + r * acc_nh[rho]);","const float nn = acc_nx[rho] + r * acc_nh[rho];")
    if e=="noxload":
        k=k.replace("if (step + 1 < L) {             // prefetch x_{t+1}","if (step + 1 < L && n == 12345) {             // prefetch x_{t+1}")
s=s[:a]+k+s[b:]
open(p,'w').write(s)
PY
}
fresh
run baseline
edit nostore; run nostore; fresh
edit nogates; run nogates; fresh
edit noxload; run noxload; fresh
edit nostore nogates noxload; run nostore_nogates_noxload; fresh
true
