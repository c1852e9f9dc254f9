cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/turboae_amd/lib/variants
for i in 1 2 3; do
  echo "base : $(python tools/quick_bench.py 50000 2>&1 | grep dec_kernel)"
  echo "prio1: $(TAE_LIB=$V/libtae_prio1.so python tools/quick_bench.py 50000 2>&1 | grep dec_kernel)"
  echo "prio2: $(TAE_LIB=$V/libtae_prio2.so python tools/quick_bench.py 50000 2>&1 | grep dec_kernel)"
done
