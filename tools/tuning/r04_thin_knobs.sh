for w in 0 1 2; do echo "== W4=$w"; WSR_CT3_W4=$w python tools/bench_conv.py thin 2>/dev/null; done
