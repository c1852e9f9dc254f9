# x_dec hashes of every kernel that writes or reads the layer-0 outputs (Y0) of the f16x2 recurrent stacks - a pure re-layout of Y0 in HBM
# must leave every one of them unchanged.  bash tools/lab/y0_layout_hashes.sh <tag>  ->  gpurun_out/y0_hashes_<tag>.txt
cd ${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p gpurun_out; out=gpurun_out/y0_hashes_$1.txt; : > $out
export TAE_DEBUG_KNOBS=1
run() { echo "[$1] $(env $1 timeout 300 python tools/lab/quick_bench_any.py ${@:2} 2>&1 | grep -v amdgpu.ids | tail -n 1)" | tee -a $out; }
R=decoder=TurboAE_rate3_rnn
# GRU decoder: both layer-0 writers x both layer-1 readers, ragged batches, short / long blocks
for B in 16384 16400 12288 500 37 16 1; do run X=0 $B $R; done
run TAE_GRU_L0=block 500 $R; run TAE_GRU_L0=unit 16384 $R
run TAE_GRU_L1=split 16384 $R; run TAE_GRU_L1=split 37 $R block_len=40
run X=0 333 $R block_len=40; run X=0 50 $R block_len=1000; run X=0 777 $R block_len=3 num_iteration=2
run X=0 2048 $R dec_num_unit=64; run X=0 100 $R dec_num_unit=37 num_iter_ft=3
# LSTM / vanilla RNN decoders: fused and split layer 1, one- and two-tile layer 0
for cell in lstm rnn; do
  for B in 16384 2100 2048 500 37 1; do run X=0 $B $R dec_rnn=$cell; done
  run TAE_RNN_L1=split 16384 $R dec_rnn=$cell; run TAE_RNN_L1=fused 500 $R dec_rnn=$cell; run TAE_RNN_L1=fused 37 $R dec_rnn=$cell block_len=40
  run X=0 300 $R dec_rnn=$cell block_len=1000; run X=0 200 $R dec_rnn=$cell dec_num_unit=48 block_len=64
done
# recurrent encoders (2 layers) in front of recurrent decoders
E2=encoder=TurboAE_rate3_rnn
for ec in gru lstm rnn; do for dc in gru lstm; do run X=0 4096 $E2 $R enc_rnn=$ec dec_rnn=$dc; run X=0 37 $E2 $R enc_rnn=$ec dec_rnn=$dc block_len=40 enc_num_unit=64; done; done
echo "== $(grep -c sha $out) hashes in $out"
