"""NaiveBWTSearcher restated (TEST INFRASTRUCTURE ONLY).

Reference: src/main/scala/org/fmindex/findex.scala:459-506 -- the searcher
BWTMerger2.calcGaps uses over one block's raw BWT: the inverted list skips row
`rk0` (the block's EOF row), `occ` masks c & 0xff, and a bucket whose last slot
is the never-written hole reports (iend - istart) (:500-502).
"""
import numpy as np


class NaiveBWTSearcher:
    def __init__(self, bwt, bucketStarts, rk0):
        self.bwt = np.asarray(bwt, dtype=np.uint8)
        self.bucketStarts = [int(x) for x in bucketStarts]
        self.K = len(self.bucketStarts)
        self.n = int(self.bwt.size)
        oct_ = [0] * self.n                      # findex.scala:463-477
        bkt = list(self.bucketStarts)
        for i in range(self.n):
            c = int(self.bwt[i]) & 0xFF
            j = bkt[c]
            if i != rk0:
                oct_[j] = i
                bkt[c] = j + 1
        self.occtable = oct_

    def cf(self, c):
        return self.bucketStarts[c]

    def occ(self, c, key):                       # findex.scala:479-505
        ci = c & 0xFF
        istart = self.bucketStarts[ci]
        imin = istart
        iend = self.n - 1 if ci == self.K - 1 else self.bucketStarts[ci + 1] - 1
        imax = iend
        if imin <= imax:
            found = False
            imid = 0
            ival = 0
            while not found and imax >= imin:
                imid = (imax + imin) // 2
                ival = self.occtable[imid]
                if ival < key:
                    imin = imid + 1
                elif ival > key:
                    imax = imid - 1
                else:
                    found = True
            if imid == iend and ival == 0:
                return iend - istart
            if ival <= key:
                return imid - istart + 1
            return imid - istart
        return 0
