#!/bin/bash
# timing-only experiments on the backbone kernel (results are wrong by construction); restores the source afterwards
cd /root/repo
cp svdd_amd/csrc/svdd_nets.hip /tmp/nets_orig.hip
run() { make -C svdd_amd/csrc 2>&1 | grep -E " error" ; echo "$1: $(timeout 120 python tools/backbone_microbench.py 256 200 | grep one-launch)"; }
edit() { python3 - "$@" <<'PY'
import sys
p='svdd_amd/csrc/svdd_nets.hip'
s=open(p).read()
a=s.index("template <bool SPT1>\n__global__ __launch_bounds__(512, 2) void backbone_kernel")
k=s[a:]
for e in sys.argv[1:]:
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
run baseline
for v in nobranch; do edit $v; run $v; cp /tmp/nets_orig.hip svdd_amd/csrc/svdd_nets.hip; done
edit noA noB noLN nobranch; run all4; cp /tmp/nets_orig.hip svdd_amd/csrc/svdd_nets.hip
make -C svdd_amd/csrc 2>&1 | grep -E " error"
true
