timeout 300 python tools/stack_stamps.py 16 1000 2>&1 | sed -n 1,4p | cut -c1-150
