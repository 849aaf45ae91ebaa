python -m pytest tests/test_refine_gpu.py tests/test_refine_golden_gpu.py tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -3
for v in 0 1; do echo "G4D_PE_L1_MFMA=$v"; G4D_PE_L1_MFMA=$v python scripts/time_pe.py 2>&1 | grep -v amdgpu; G4D_PE_L1_MFMA=$v python scripts/time_model.py 8 30 8192 5 2>&1 | grep "forward"; done
