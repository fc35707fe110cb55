# same-box A/B of one FCL_<NAME> tunable on the two training updates: tools/tunable_ab.sh <out> <NAME> <value A> <value B>
OUT=gpurun_out/${1:-tab}; NAME=$2; A=$3; B=$4
mkdir -p $OUT
val() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'])"; }
for rep in 1 2 3; do
 for w in kd_step teacher_step; do
  a=$(env FCL_$NAME=$A python3 bench.py --workload $w --no-cpu-baseline --no-dp-schedule --regions 5 2>>$OUT/err.log | val)
  b=$(env FCL_$NAME=$B python3 bench.py --workload $w --no-cpu-baseline --no-dp-schedule --regions 5 2>>$OUT/err.log | val)
  echo "$w rep $rep FCL_$NAME=$A $a  FCL_$NAME=$B $b" >> $OUT/ab.log
 done
done
cat $OUT/ab.log
