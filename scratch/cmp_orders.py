import sys, numpy as np
a = np.load(sys.argv[1]); b = np.load(sys.argv[2])
bad = [k for k in a.files if not np.array_equal(a[k], b[k])]
print(len(a.files), "clouds; differing:", bad)
sys.exit(1 if bad else 0)
