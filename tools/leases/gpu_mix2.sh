timeout 900 python tools/time_mix.py 5000000 256 10 8 24 2>&1 | tail -24
