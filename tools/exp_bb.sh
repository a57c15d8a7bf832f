#!/bin/bash
# timing-only experiments on the backbone kernel (results are wrong by construction)
# The tracked sources are never touched: the kernels are copied to a scratch directory, patched and built THERE, and the
# microbenchmark loads that build through SVDD_HIP_LIB (svdd_amd/_lib.py). The scratch directory is removed on any exit.
cd "$(dirname "$0")/.." || exit 1
ROOT=$PWD
WORK=$(mktemp -d /tmp/svdd_exp.XXXXXX)
trap 'rm -rf "$WORK"' EXIT
fresh() { cp svdd_amd/csrc/*.hip svdd_amd/csrc/Makefile "$WORK"/; }
build() { make -C "$WORK" -j3 INC="$ROOT/include" 2>&1 | grep -E " error"; }
run() { build; echo "$1: $(SVDD_HIP_LIB=$WORK/libsvdd_hip.so timeout 120 python tools/backbone_microbench.py 256 200 | grep one-launch)"; }
edit() { python3 - "$WORK" "$@" <<'PY'
import sys
p=sys.argv[1]+'/svdd_nets.hip'
s=open(p).read()
a=s.index("template <bool SPT1>\n__global__ __launch_bounds__(512, 2) void backbone_kernel")
k=s[a:]
for e in sys.argv[2:]:
    if e=="noA":
        assert "        V[0] = ap_[0]; V[1] = ap_[1]; }" in k
        k=k.replace("        V[0] = ap_[0]; V[1] = ap_[1]; }","        V[0] = float4{bf0[0],bf0[1],bf0[2],bf0[3]}; V[1] = V[0]; (void)ap_; }",1)
    if e=="noB":
        assert "      if (nxt < it_end) {\n        const float* src = wsrc + (size_t)tile_of(nxt) * BB_C * CH;" in k
        k=k.replace("      if (nxt < it_end) {\n        const float* src = wsrc + (size_t)tile_of(nxt) * BB_C * CH;","      if (nxt < it_end && a.n == 12345) {\n        const float* src = wsrc + (size_t)tile_of(nxt) * BB_C * CH;",1)
    if e=="nobranch":
        assert "      if (live & (1 << (2 * (R)))) {" in k
        k=k.replace("      if (live & (1 << (2 * (R)))) {","      if (true) {",1)
    if e=="nofence":
        k=k.replace("      __builtin_amdgcn_sched_barrier(0);                                                                     \\\n      B2_WAIT(NOUT)","      /*nofence*/")
    if e=="noLN":
        assert "    if (layer < nl) {\n      const float tb0 = vl[BB_C + col0]" in k
        k=k.replace("    if (layer < nl) {\n      const float tb0 = vl[BB_C + col0]","    if (layer < nl && a.n == 12345) {\n      const float tb0 = vl[BB_C + col0]",1)
s=s[:a]+k
open(p,'w').write(s)
PY
}
fresh
run baseline
for v in nobranch; do edit $v; run $v; fresh; done
edit noA noB noLN nobranch; run all4; fresh
true
