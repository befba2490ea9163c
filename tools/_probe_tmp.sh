mkdir -p gpurun_out/r4i && export TMPDIR=/tmp
python3 tools/bench_configs.py --only c5 --gen both > gpurun_out/r4i/configs_c5.jsonl 2> gpurun_out/r4i/configs_c5.err || exit 1
cd gpurun_out/r4i
for sh in "arxiv --heads 1" "arxiv --heads 8" "arxiv --heads 4" "arxiv --heads 2" "arxiv --heads 1 --width 64" "arxiv-rmat --heads 1" "arxiv-rmat --heads 8" "rmat20 --heads 1" "rmat20 --heads 4" "rmat20 --heads 8"; do
  python3 ../../tools/gat_probe.py $sh >> p_default.jsonl 2>probe.err || exit 1
  SGX_GAT_FUSED=0 python3 ../../tools/gat_probe.py $sh >> p_two_stage.jsonl 2>probe.err || exit 1
  SGX_GAT_FUSED=0 SGX_GAT_SCAN=0 python3 ../../tools/gat_probe.py $sh >> p_two_stage_rows.jsonl 2>probe.err || exit 1
done
for sh in "arxiv --heads 8" "rmat20 --heads 1" "rmat20 --heads 8" "arxiv-rmat --heads 8"; do
  tag=$(echo $sh | tr -d ' -')
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o t -- python3 ../../tools/gat_probe.py $sh --launches 10 > /dev/null 2>prof_$tag.err || exit 1
  cp $(find /tmp/prof_$tag -name '*kernel_stats.csv' | head -1) kernel_stats_$tag.csv
done
python3 - <<'P'
import json
a=[json.loads(l) for l in open("p_default.jsonl")]; b=[json.loads(l) for l in open("p_two_stage.jsonl")]; c=[json.loads(l) for l in open("p_two_stage_rows.jsonl")]
for x,y,z in zip(a,b,c): print(x["shape"], x["heads"], x["width"], "default", x["ms_gat_aggregate"], "two-stage", y["ms_gat_aggregate"], "rows-form", z["ms_gat_aggregate"], "plain", x["ms_plain_aggregate"])
P
