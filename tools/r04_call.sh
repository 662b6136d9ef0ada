mkdir -p gpurun_out/r04f2; O=gpurun_out/r04f2
(timeout 1500 python tools/fuzz_parity.py 530000 3000 > $O/fuzz_a.log 2>&1; tail -1 $O/fuzz_a.log) &
(timeout 1500 python tools/fuzz_parity.py 540000 1200 48 32 8 120 > $O/fuzz_b.log 2>&1; tail -1 $O/fuzz_b.log) &
(timeout 1500 python tools/fuzz_parity.py 550000 1500 48 32 8 26 hostile > $O/fuzz_h.log 2>&1; tail -1 $O/fuzz_h.log) &
(timeout 1500 python tools/fuzz_parity.py 560000 600 96 64 16 60 > $O/fuzz_c.log 2>&1; tail -1 $O/fuzz_c.log) &
wait
for f in a b h c; do echo "$f: $(grep -c '^seed' $O/fuzz_$f.log) seeds, worst $(grep -o 'worst rel err [0-9.e+-]*' $O/fuzz_$f.log | awk '{print $4}' | sort -g | tail -1), errors $(grep -c 'ERROR\|ABOVE' $O/fuzz_$f.log)"; done
