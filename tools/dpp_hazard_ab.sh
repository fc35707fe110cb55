# Round 6: what the lane-split BiLSTM kernels' concurrency failure depends on.  Four builds of bilstm.hip (tools/ab/, built by hand: EXTRA=-DFCL_KS_NOT_EXCLUSIVE and / or
# -DFCL_KS_BPERMUTE) run tools/stress_bilstm_concurrent.py (H = 256 group kernel beside a stream of H = 128 kernels; every result against the serial one).
OUT=gpurun_out/${1:-r6g}
mkdir -p $OUT
for v in "in-tree (exclusive, DPP)|" "exclusive, ds_bpermute|tools/ab/libfcl_excl_bperm.so" "NOT exclusive, DPP|tools/ab/libfcl_ksne_dpp.so" "NOT exclusive, ds_bpermute|tools/ab/libfcl_ksne_bperm.so"; do
  name=${v%%|*}; lib=${v##*|}
  for order in 0 1; do
    if [ -n "$lib" ]; then export FCL_LIB=$PWD/$lib; else unset FCL_LIB; fi
    r=$(FCL_KS_GUARD=0 STRESS_ITERS=150 STRESS_SMALL_FIRST=$order timeout 300 python3 tools/stress_bilstm_concurrent.py 2>&1 | grep -v -i "warn\|amdgpu.ids" | tail -1)
    echo "$name | small-first=$order | $r" >> $OUT/dpp_hazard_ab.log
  done
done
unset FCL_LIB
cat $OUT/dpp_hazard_ab.log
