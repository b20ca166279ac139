cd ${GRAFT_REPO_ROOT:-$PWD}
export BGAMD_REHEARSE_ONE_GPU=1
# (1) the driver's launch line; (2) the plain launch: bench.py starts its own ranks
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --height 384 --width 256 --batch 4 > gpurun_out/ddp2.json 2> gpurun_out/ddp2.err; echo rc=$?
grep "\[bench\]" gpurun_out/ddp2.err | tail -4; cat gpurun_out/ddp2.json | cut -c1-300
timeout -k 10 400 python bench.py --gpus 2 --steps 3 --warmup 1 --height 384 --width 256 --batch 4 > gpurun_out/ddp2_plain.json 2> gpurun_out/ddp2_plain.err; echo rc=$?
grep "\[bench\]" gpurun_out/ddp2_plain.err | head -3; cat gpurun_out/ddp2_plain.json | cut -c1-300
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --height 384 --width 256 --batch 4 --no-kernel-profile --no-cpu-baseline 2>/dev/null | cut -c1-200
