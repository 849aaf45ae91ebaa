mkdir -p gpurun_out
python -m pytest tests/test_ops_gpu.py tests/test_contraction.py -x -q -m gpu -k "fps or FPS or contraction" 2>&1 | tail -5 > gpurun_out/r5_fps_tests.log
for soa in 1 0; do echo "SOA=$soa" >> gpurun_out/r5_fps_time.log; G4D_FPS_SOA=$soa G4D_FPS_BUCKET_W=16 G4D_FPS_DEAL=8 python scripts/time_fps.py run >> gpurun_out/r5_fps_time.log 2>&1; done
for soa in 1 0; do echo "SOA=$soa" >> gpurun_out/r5_overlap.log; G4D_FPS_SOA=$soa PAIRS="fps_gather_grid:group[mlp_chain_group_table+mlp_chain_group_table]#1,fps_gather_grid:mlp_chain_interp_init,fps_gather_grid:linear#3,fps_gather_grid:linear_interp_add,fps_gather_grid:sa_xyz,fps_gather_grid:mlp_chain_table_cells,fps_gather_grid:lbs_one,fps_gather_grid:three_nn" python scripts/exp_overlap.py 240 >> gpurun_out/r5_overlap.log 2>&1; done
cat gpurun_out/r5_fps_tests.log gpurun_out/r5_fps_time.log; grep -v "^alone" gpurun_out/r5_overlap.log
