import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(),'tests'))
import numpy as np
from conftest import fixture_reads, pack_reads
from oracle import pyorc
import hulk_amd
reads = fixture_reads()
k,w=21,9
for nr in (1,2,5,1000):
    g = hulk_amd.GpuSketcher(k,w,4)
    b,o = pack_reads(reads[:nr]); g.add_reads(b,o)
    gh = g.histogram()
    oh = np.zeros(k**4, dtype=np.uint32)
    for r in reads[:nr]:
        for x in pyorc.minimizers(r,k,w): oh[pyorc.jump(int(x), k**4)] += 1
    d = np.nonzero(gh!=oh)[0]
    print(nr, 'gpu sum', gh.sum(), 'orc sum', oh.sum(), 'diff bins', len(d), g.counters())
    if nr==1:
        print(' gpu bins', np.nonzero(gh)[0][:40], gh[np.nonzero(gh)[0]][:40])
        print(' orc bins', np.nonzero(oh)[0][:40])
    g.close()
