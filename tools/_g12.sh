mkdir -p gpurun_out/r2k
(timeout 600 python tools/prof_seq.py 2>&1 | tail -60) > gpurun_out/r2k/prof224.log
(timeout 600 python tools/prof_seq.py --size 512 --frames 50 --train-policy 2>&1 | tail -90) > gpurun_out/r2k/prof512.log
