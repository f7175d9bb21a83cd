import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
from webgraph_amd import bvgraph as B
from oracle import oracle as O
g = O.OracleGraph.load("/root/repo/tests/golden/cnr-2000")
rowptr, succ, _ = g.scan()
try:
    B.compress(rowptr, succ, windowSize=7, maxRefCount=3, minIntervalLength=3, zetaK=3)
except Exception as e:
    print("ERR", e)
