python -m pytest tests/test_learner.py tests/test_device_replay.py tests/test_reference_callers.py -x -q -m gpu > gpurun_out/r4_t20.log 2>&1; tail -8 gpurun_out/r4_t20.log
python tools/learner_profile.py --fused 2>/dev/null
python tools/learner_profile.py --fused --game Hanabi-Small 2>/dev/null
python tools/loop_bench.py 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(d['learner_steps_per_s'], d['host_ms_per_learner_step_enqueue'], d['host_wait_for_the_gpu_ms_per_learner_step'], d['loss_last'])"
