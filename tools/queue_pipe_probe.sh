#!/bin/bash
# Hardware-queue / pipe placement probes (profiles/r06_queue_pipes.txt). Run on the GPU box from the repository root.
#  1. the batch receiver's overlap mode at one frame per call, the caller's stream on the 1st .. 4th hardware queue of the process
#     (idle streams made ahead of it) and on the legacy NULL stream: flat since the handle runs the call on four streams of its own
#  2. the slot-shaped path with 0 .. 3 idle hardware queues made ahead of the LDPC stage's submit stream (T2GPU_LDPC_QUEUE_SKIP): the
#     decode's queue moves over the four pipes of the command processor; two of the four positions share a pipe with the per-symbol chain's
#     or the equaliser's queue
set -u
for k in 0 1 2 3; do
    echo -n "rx overlap, 1 frame per call, host end on, caller's stream on hardware queue $((k + 1)): "
    PROBE_STREAM_FIRST=1 PROBE_DUMMY=$k python tools/overlap_call_probe.py 1 --ts 2>&1 | grep Msamples | cut -d';' -f1
done
echo -n "rx overlap, 1 frame per call, host end on, caller on the NULL stream: "
PROBE_NULL_STREAM=1 python tools/overlap_call_probe.py 1 --ts 2>&1 | grep Msamples | cut -d';' -f1
for sk in 0 1 2 3; do
    for r in 1 2; do
        echo -n "drop-in, $sk idle queue(s) ahead of the decodes' queue: "
        T2GPU_LDPC_QUEUE_SKIP=$sk python bench.py --only-drop-in 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'], 'Msamples/s')"
    done
done
