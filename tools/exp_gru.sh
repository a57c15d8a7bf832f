#!/bin/bash
# timing-only experiments on gru_bidir_kernel (results wrong by construction); restores the source afterwards
cd /root/repo
cp svdd_amd/csrc/svdd_nets.hip /tmp/nets_orig.hip
run() { make -C svdd_amd/csrc 2>&1 | grep -E " error" ; echo "$1: $(timeout 120 python tools/gru_microbench.py 16 2048 2560 2>&1 | grep 'mode=0' | tr '\n' ' ')"; }
edit() { python3 - "$@" <<'PY'
import sys
p='svdd_amd/csrc/svdd_nets.hip'
s=open(p).read()
a=s.index("template <bool BOTH>"); b=s.index("// ------------------------------------------------------------------ fused conv epilogue + LayerNorm ----")
k=s[a:b]
for e in sys.argv[1:]:
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
run baseline
edit nostore; run nostore; cp /tmp/nets_orig.hip svdd_amd/csrc/svdd_nets.hip
edit nogates; run nogates; cp /tmp/nets_orig.hip svdd_amd/csrc/svdd_nets.hip
edit noxload; run noxload; cp /tmp/nets_orig.hip svdd_amd/csrc/svdd_nets.hip
edit nostore nogates noxload; run nostore_nogates_noxload; cp /tmp/nets_orig.hip svdd_amd/csrc/svdd_nets.hip
make -C svdd_amd/csrc 2>&1 | grep -E " error"
true
