cd $GRAFT_REPO_ROOT
O=gpurun_out/r3h; mkdir -p $O
timeout 1700 python -m pytest tests/test_gpu_engine.py tests/test_gpu_fullsize.py tests/test_gpu_sharded_encode.py -q -s -k "row_bucketed or int8_weight_mode_matches or full_depth_13b or full_depth_codellama or outlier_feature or sharded or add_schemas or library" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -n "full depth\|product vs oracle\|\[int8\|passed\|failed\|Error\|rc=\|FAILED" $O/pytest.log | tail -40
