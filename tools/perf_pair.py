#!/usr/bin/env python3
"""Device-resident throughput of one rate pair: perf_pair.py in_rate out_rate [channels] [streams]"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
import perf_matrix as pm
fi, fo = int(sys.argv[1]), int(sys.argv[2])
nch = int(sys.argv[3]) if len(sys.argv) > 3 else 2
S = int(sys.argv[4]) if len(sys.argv) > 4 else 256
print(json.dumps(pm.run("%d->%d %dch" % (fi, fo, nch), fi, fo, nch, S, {})))
