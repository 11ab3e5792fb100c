#!/bin/bash
python bench.py --no-cpu-baseline --no-others --no-large > /tmp/out.txt 2>/tmp/err.txt; echo "stdout lines: $(wc -l < /tmp/out.txt)"; head -c 300 /tmp/out.txt; echo
BOSSX_FORCE_COLLECTIVES=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29573 python bench.py --no-cpu-baseline --no-others --no-large > /tmp/out2.txt 2>/tmp/err2.txt; echo "stdout lines (forced collectives): $(wc -l < /tmp/out2.txt)"; python3 -c "
import json
d=json.loads(open('/tmp/out2.txt').read())
print('forced: ms_per_step %.3f collectives %s' % (d['ms_per_step'], d['config']['collectives_per_update']))"
tail -2 /tmp/err2.txt | cut -c1-200
