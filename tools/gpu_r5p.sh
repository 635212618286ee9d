#!/bin/bash
# round 5, GPU session P: the N > 1 code path in the driver's launcher form at world size 1 (RCCL), default / serial collective / padded gather
D=gpurun_out/r5p
mkdir -p $D
L="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517"
FS2_FORCE_DIST=1 timeout 300 $L bench.py --gpus 1 --steps 20 --warmup 5 --workload c5 > $D/launcher_default.json 2>$D/launcher_default.err
FS2_FORCE_DIST=1 FS2_DIST_SERIAL=1 timeout 300 $L bench.py --gpus 1 --steps 20 --warmup 5 --workload c5 > $D/launcher_serial.json 2>$D/launcher_serial.err
FS2_FORCE_DIST=1 timeout 300 $L bench.py --gpus 1 --steps 20 --warmup 5 --workload c5 --padded > $D/launcher_padded.json 2>$D/launcher_padded.err
timeout 300 $L bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $D/launcher_c3.json 2>$D/launcher_c3.err
for f in $D/*.json; do echo "$f $(tail -n 1 $f | python -c "import json,sys;d=json.loads(sys.stdin.read());print(d['value'], d['ms_per_step'], d.get('sustained_ms_per_step'), d['config'].get('streams'), d['config'].get('gather'), (d.get('one_stream') or {}).get('value'), (d.get('multi_gpu') or {}).get('ms_per_step_serial_collective'))" 2>&1 | tail -1)"; done
tail -n 2 $D/launcher_default.err
