"""mz_preyama_batch() from block text to block rows: rate, link bytes, a sample against the reference's pre_yama().
    python tests/tools/prepath.py [config] [pairs/0] [check]        (MZ_TIMING=1|2: the library's JSON lines per call / per chunk)"""
import json, os, sys
sys.path.insert(0, '.')
import bench
from multiz_amd import api, synth
api.init(0)
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
pairs = int(sys.argv[2]) if len(sys.argv) > 2 and int(sys.argv[2]) else synth.CONFIGS[cfg]["pairs"]
check = int(sys.argv[3]) if len(sys.argv) > 3 else 200
print(json.dumps(bench.pre_column(cfg, pairs, check)))
