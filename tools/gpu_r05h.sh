#!/usr/bin/env bash
# round 5: wide decoders' weight gradients as split-bf16 products; the ecd family as shipped; a timeline of a TIMED pixel step
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_train.py "tests/test_gpu_frontend.py::test_pixel_regime_defaults_match_the_oracle" -q -x > $OUT/r05h_pytest.log 2>&1
echo "pytest exit $?"; tail -4 $OUT/r05h_pytest.log
for b in 32 128; do
  timeout 300 python bench.py --regime landmarks_attn --model lstm768 --attention none --char-dim 256 --rnn-dropout 0.3 --batch $b --no-cpu-baseline 2>$OUT/r05h_ecd$b.err | tail -1 > $OUT/r05h_ecd$b.json
  python -c "
import json; d=json.load(open('$OUT/r05h_ecd$b.json')); print('ecd B=$b', d['ms_per_step'], d['config'].get('decoder_recurrence'))" || tail -3 $OUT/r05h_ecd$b.err
done
bash tools/gpu_kt.sh r05h_ecd32 --regime landmarks_attn --model lstm768 --attention none --char-dim 256 --batch 32 > /dev/null
head -12 $OUT/r05h_ecd32_kernel_stats.txt | cut -c1-140
bash tools/gpu_timeline.sh r05h_px conv1_fwd --regime pixels > /dev/null
cat $OUT/r05h_px_step_timeline.txt | cut -c1-110
