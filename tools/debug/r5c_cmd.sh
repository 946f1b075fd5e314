for L in hip vbwarm vbnoslow vbnoflush; do
  export CASMVS_LIB_PATH=$GRAFT_REPO_ROOT/casmvsnet_pl_amd/libcasmvs_$L.so
  mkdir -p $OUT/$L
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$L/prof -o stats -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 10 --warmup 3 > $OUT/$L/bench.json 2> $OUT/$L/bench.err )
  find $OUT/$L/prof -name "*.db" -delete; find $OUT/$L/prof -type f -size +4M -delete
  echo "== $L: $(grep -o '"train_step_ms": [0-9.]*' $OUT/$L/bench.json)"
  f=$(find $OUT/$L/prof -name "*kernel_stats.csv" | head -1)
  grep -E "absmax|costvol_var_bwd|fixed_finish" $f | sed -E 's/\(float const.*\)",/",/' | cut -c1-160
done
