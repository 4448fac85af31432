export TMPDIR=/tmp
mkdir -p gpurun_out/r6e
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_dist_gpu.py -x -q -k "bench or dist or rccl or nccl or allgather or two_ranks" > gpurun_out/r6e/pytest_dist.log 2>&1
tail -12 gpurun_out/r6e/pytest_dist.log
