"""Idle-gap analysis of a rocprofv3 kernel trace of replayed bench steps: takes the last replayed step (the kernels
between the last two long idle periods), reports wall time, the union of kernel intervals (GPU busy), time with two or
more kernels in flight, the number of kernels, and the distribution of idle gaps between consecutive kernels."""
import csv
import sys


def main(path):
  rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(path))]
  rows.sort()
  # steps are separated by host-side syncs: find gaps > 200 us
  cuts = [0]
  end = rows[0][1]
  for i in range(1, len(rows)):
    if rows[i][0] - end > 200_000:
      cuts.append(i)
    end = max(end, rows[i][1])
  cuts.append(len(rows))
  segs = [(cuts[i], cuts[i + 1]) for i in range(len(cuts) - 1)]
  # the timed region: the longest segments by kernel count at the end
  big = [s for s in segs if s[1] - s[0] > 500]
  print('segments with > 500 kernels:', [(b - a) for a, b in big][-6:])
  a, b = big[-1]
  seg = rows[a:b]
  t0, t1 = seg[0][0], max(r[1] for r in seg)
  ev = []
  for s, e, _ in seg:
    ev.append((s, 1))
    ev.append((e, -1))
  ev.sort()
  busy = two = 0
  depth = 0
  last = ev[0][0]
  for t, d in ev:
    if depth >= 1:
      busy += t - last
    if depth >= 2:
      two += t - last
    depth += d
    last = t
  print('kernels %d  wall %.3f ms  busy(union) %.3f ms  idle %.3f ms  >=2 in flight %.3f ms  sum of durations %.3f ms' % (
      len(seg), (t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, two / 1e6, sum(e - s for s, e, _ in seg) / 1e6))
  # idle gaps of the union
  gaps = []
  end = seg[0][1]
  for s, e, name in seg[1:]:
    if s > end:
      gaps.append(((s - end) / 1e3, name))
    end = max(end, e)
  gaps.sort(reverse=True)
  import collections
  hist = collections.Counter()
  for gval, _ in gaps:
    hist[min(int(gval), 20)] += 1
  print('idle gaps: n=%d total %.3f ms; histogram by us (20 = >=20):' % (len(gaps), sum(g for g, _ in gaps) / 1e3), sorted(hist.items()))
  print('largest gaps (us, kernel that follows):')
  for gval, name in gaps[:8]:
    print('  %.1f  %s' % (gval, name[:100]))
  fam = collections.defaultdict(lambda: [0, 0])
  for s_, e_, name in seg:
    key = (name[5:] if name.startswith('void ') else name).replace('(anonymous namespace)::', '').split('(')[0][:70]
    fam[key][0] += (e_ - s_) / 1e3
    fam[key][1] += 1
  if len(sys.argv) > 2:      # per-kernel table of this ONE replayed step (a rocprofv3 --stats file also counts the eager warm-up)
    import json
    with open(sys.argv[2], 'w') as fh:
      json.dump({'kernels': len(seg), 'wall_ms': (t1 - t0) / 1e6, 'busy_ms': busy / 1e6, 'two_in_flight_ms': two / 1e6,
                 'sum_ms': sum(e - s for s, e, _ in seg) / 1e6,
                 'by_kernel': sorted(([k, v[1], round(v[0], 1)] for k, v in fam.items()), key=lambda r: -r[2])}, fh, indent=1)
  # the library's kernels all live in anonymous namespaces (mangled: _GLOBAL__N_1; demangled: the prefix `key` dropped);
  # what the framework or the runtime launches carries its own namespace (at::native::, __amd_rocclr_, rccl)
  print('kernels that are not this library\'s (framework / runtime launches left in the step):')
  foreign = [(k, v) for k, v in fam.items() if k.startswith(('at::', '__amd_', 'rccl', 'nccl', 'hip'))]
  for k, v in sorted(foreign, key=lambda kv: -kv[1][0]):
    print('  %8.1f us %4d  %s' % (v[0], v[1], k))
  print('  total: %d launches, %.1f us' % (sum(v[1] for _, v in foreign), sum(v[0] for _, v in foreign)))


if __name__ == '__main__':
  main(sys.argv[1])
