# Same-box A/B of launcher tunables on the default synthesis line: prints fresh-feed and replayed M frames/s per setting (two rounds).
# usage: bash tools/sweep_env.sh "FCL_FP_ROW_TILES=1" "FCL_FP_ROW_TILES=4" ...   ("" = defaults)
for round in 1 2; do
  for cfg in "" "$@"; do
    echo -n "[$round] ${cfg:-defaults}: "
    env $cfg python3 bench.py --no-cpu-baseline --no-extras --regions 11 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), round(d.get('value_replay_only',0)/1e6,2), round(d['roofline']['frac'],4))"
  done
done
