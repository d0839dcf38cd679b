for w in 5488 5490 5492; do python tools/units_probe.py w=$w 6 2>&1 | grep -v "one by one" | tail -2; done
