"""development: batch sizes of the shared launches from a LPSLAM_HIP_SHARE_TRACE=1 log (steady-state part of the last run)"""
import re, collections, sys
L = [l for l in open(sys.argv[1]) if l.startswith('share ')]
ev = [(float(l.split()[1]), l.split()[2], int(l.split()[3]), l) for l in L]
tend = ev[-1][0]
span = float(sys.argv[2]) if len(sys.argv) > 2 else 200
ph = [e for e in ev if tend - span - 15 < e[0] < tend - 15]
for kind in ('pose', 'proj', 'front', 'solve'):
    c = collections.Counter(e[2] for e in ph if e[1] == kind)
    n = sum(c.values()); tot = sum(k * v for k, v in c.items())
    print(kind, 'batches', n, 'requests', tot, 'mean %.2f' % (tot / max(n, 1)), sorted(c.items()))
fr = [e for e in ph if e[1] == 'front']
g = [float(re.search(r'gathered for (\d+)', e[3]).group(1)) for e in fr]
la = [float(re.search(r'launched in (\d+)', e[3]).group(1)) for e in fr]
if g: print('front: gather mean %.0f us, launch mean %.0f us, max %.0f' % (sum(g) / len(g), sum(la) / len(la), max(la)))
if len(sys.argv) > 3:
    t = tend - span / 2
    for e in ph:
        if t < e[0] < t + float(sys.argv[3]): print(e[3].strip())
