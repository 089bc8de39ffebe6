cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r1
timeout 1500 python -m pytest tests/test_gpu_p2p.py -x -q 2>&1 | tail -15 > gpurun_out/r1/p2p.txt
for p in f32 bf16 x2; do timeout 300 python tools/dp_probe.py $p 2>&1 | grep "^{" >> gpurun_out/r1/dp.txt; done
timeout 600 python tools/form_table.py --json gpurun_out/r1/launch_forms.json "tools/form_table.py on MI355X (256 compute units), round 6 (dp_inline_form column; B = 128 column; exporting f32 / bf16 DDPG learners keep the mirrored / uncached packs)" > gpurun_out/r1/forms.txt 2>&1
timeout 200 python tools/quick_rate.py f32 bf16 x2 > gpurun_out/r1/rate.txt 2>&1
cat gpurun_out/r1/p2p.txt gpurun_out/r1/dp.txt gpurun_out/r1/forms.txt gpurun_out/r1/rate.txt
