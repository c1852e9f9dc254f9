# A/B of the long-block segment geometry on the GPU box: equal segments of at most T centre positions (TAE_SEG_T) vs the default
cd $GRAFT_REPO_ROOT
export TAE_DEBUG_KNOBS=1      # the library ignores its debug knobs without it
for T in 125 143 167 200 250 ""; do
  echo "== TAE_SEG_T=$T"
  TAE_SEG_T=$T python tools/lab/quick_bench_cfg.py 1000 25000 2 2>&1 | grep -v amdgpu.ids
done
