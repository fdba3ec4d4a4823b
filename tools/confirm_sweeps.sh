set -u
out=gpurun_out/r6t_confirm_sweeps.txt
echo "# randomised parity sweeps on the last library of round 6 (bit-image route, sym_skinny rewrite), one line per process" > $out
i=0
for spec in "random_parity2.py 30 4001" "random_parity3.py 30 4002" "random_parity4.py 24 4003" "random_parity4.py 24 4004" "random_parity5.py 16 4005" "random_parity.py 40 4006"; do
  set -- $spec
  r=$(timeout -k 10 400 python tools/$1 $2 $3 2>&1 | grep -i "failures" | tail -1)
  echo "$1 $2 $3: $r" >> $out
done
cat $out
