#!/bin/bash
# does the fused wq|wk|wv + attention (+ wo) launch pay at 13B / 65B widths at short contexts?  (rule in fused_qkv_shape)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5g
for cfg in "13b q5_1" "65b q8_0"; do set -- $cfg
for fa in 1 2; do
GGML_HIP_FUSE_ATTN=$fa timeout 600 python bench.py --model $1 --wtype $2 --weights blocks --no-cpu-baseline --prefill-steps 0 --steps 48 > gpurun_out/r5g/b_$1_fa$fa.json 2> gpurun_out/r5g/b_$1_fa$fa.err
python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r5g/b_$1_fa$fa.json').read().strip().splitlines()[-1]); r=d['roofline']
    print('$1 $2 fuse_attn=$fa', d['value'], {k:(v['launches'], v['us_per_launch']) for k,v in r['per_kind'].items()}, d['parity_check']['passed'], d['config']['long_context']['tokens_per_s'], d['config']['decode_launches']['qkv_and_attention_in_one_launch_tokens'], d['config']['decode_launches']['wo_in_the_attention_launch_tokens'])
except Exception as e: print('$1 $2 failed', e)
PY
done; done
