#!/usr/bin/env python3
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
import perf_matrix as pm
idx = int(sys.argv[1])
print(json.dumps(pm.run(*pm.CASES[idx])))
