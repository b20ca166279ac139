cd ${GRAFT_REPO_ROOT:-$PWD}
for t in 512 384 256 1024; do echo "== WGRAD_TARGET=$t"; BGAMD_WGRAD_TARGET=$t LDPAD=32 timeout -k 10 200 python scripts/bench_conv.py 2>/dev/null | grep -E "728->  728|256->  256 k3|1536|128->  128 k1 s1 d 1  576|weighted wgrad"; done
