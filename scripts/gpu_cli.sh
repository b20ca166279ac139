set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/cli /tmp/cli
timeout -k 10 300 python -m bias_gan_amd.gpsro_train.train_gan --channels 0 1 2 3 --synthetic_size 64 96 --local_batch_size 2 --max_steps 6 --logging_frequency 2 --save_frequency 3 --output_dir /tmp/cli --lr_schedule_generator type=multistep,milestones=2\ 4,decay_rate=0.5 > gpurun_out/cli/run1.log 2>&1
timeout -k 10 300 python -m bias_gan_amd.gpsro_train.train_gan --channels 0 1 2 3 --synthetic_size 64 96 --local_batch_size 2 --max_steps 8 --logging_frequency 1 --checkpoint /tmp/cli/gan_step_6.cpt --output_dir /tmp/cli --amp_opt_level O0 > gpurun_out/cli/run2.log 2>&1
for f in gpurun_out/cli/run1.log gpurun_out/cli/run2.log; do echo == $f; tail -n 8 $f; done
