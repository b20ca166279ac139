cd ${GRAFT_REPO_ROOT:-$PWD}
echo "== baseline"; LDPAD=32 timeout -k 10 200 python scripts/bench_conv.py 2>/dev/null | grep -E "728->  728|256->  256 k3|2048|weighted fwd|weighted dgrad"
for v in NO_MMA NO_LOAD HOT_LOAD; do echo "== $v"; BGAMD_LIB=$PWD/abl_build/libbgamd_$v.so LDPAD=32 timeout -k 10 200 python scripts/bench_conv.py 2>/dev/null | grep -E "728->  728|256->  256 k3|2048|weighted fwd|weighted dgrad"; done
