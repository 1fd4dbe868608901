#!/usr/bin/env python3
"""Timeline of ONE fused tree + trunk launch from GAZ_FUSED_STAMPS=<file>[:n] (engine.hip; wall clock, 100 MHz): when the tree waves
finish, when the trunk workgroups start waiting, start computing and end.  usage: fused_timeline.py <file> <n_tree_blocks>"""
import sys
import numpy as np
raw = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 128).astype(np.int64)
n_tree = int(sys.argv[2])
tree, trunk = raw[:n_tree], raw[n_tree:]
trunk = trunk[trunk[:, 0] > 0]
t0 = min(tree[:, :4][tree[:, :4] > 0].min(), trunk[:, 0].min())
us = lambda x: (x - t0) / 100.0
tw_s, tw_e = us(tree[:, 0:4]).ravel(), us(tree[:, 4:8]).ravel()
pct = lambda v: " ".join(f"p{q}={np.percentile(v, q):.1f}" for q in (1, 10, 50, 90, 99, 100))
print(f"tree waves: {tw_s.size}; start {pct(tw_s)}")
print(f"            end   {pct(tw_e)}    duration {pct(tw_e - tw_s)}")
blk_end = us(tree[:, 4:8]).max(axis=1)
print(f"tree BLOCKS (slot held until the slowest of 16 games): end {pct(blk_end)}")
s, w, e = us(trunk[:, 0]), us(trunk[:, 1]), us(trunk[:, 63])
print(f"trunk workgroups: {len(trunk)}; dispatched {pct(s)}")
print(f"   wait for their games (stamp 1 - stamp 0): {pct(w - s)}   -> slot-time lost waiting: {np.sum(w - s):.0f} us over all workgroups")
print(f"   compute (stamp 63 - stamp 1): {pct(e - w)}")
print(f"   end {pct(e)}   launch span {e.max():.1f} us")
order = np.argsort(s)
for i in range(0, len(trunk), 512):
    r = order[i:i + 512]
    print(f"   dispatch round {i // 512}: {len(r)} wgs, dispatched {s[r].mean():.1f}, started {w[r].mean():.1f}, ended {e[r].mean():.1f}, compute {np.mean(e[r] - w[r]):.1f} us")
busy = np.sum(e - w)
print(f"trunk compute slot-time {busy:.0f} us = {busy / 512:.1f} us per slot; launch {e.max():.1f} us -> slot utilisation {busy / 512 / e.max():.2f}")
