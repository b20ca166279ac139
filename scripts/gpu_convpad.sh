cd ${GRAFT_REPO_ROOT:-$PWD}
for p in 0 32 64; do echo "== LDPAD=$p"; LDPAD=$p timeout -k 10 200 python scripts/bench_conv.py 2>/dev/null | grep -E "728|weighted"; done
