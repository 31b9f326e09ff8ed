#!/usr/bin/env python3
"""patches the device assembly hipcc wrote for place.hip (-save-temps) so that k_near_tie_runs, behind
its last store and with every instruction in front of it untouched, leaves 32 words of what its wave holds -- and of what it
reads when it asks again -- per lane at run_start[1024 + 32 e ..]:
  0-1 the per-lane mask "e + k <= n" (s[18:19])   2-3 packed-text pointer (s[10:11])   4-5 n   6-7 k   8-9 far-list pointer
  10 pe   11 the run start it stored   12 v16   13-14 EXEC   15 tag
  16 far[lo - 1] read again, plain load    17 the same with sc0 sc1 (system-coherent: past the caches)
  18-19 pk word of far[lo - 1] (plain, from the plain value)   20-21 the same word, sc0 sc1
  22-23 pk word of pe, plain   24-25 sc0 sc1
usage: patch_runs_asm.py in.s out.s"""
import re
import sys

src = open(sys.argv[1]).read().split("\n")
out, inside, done = [], False, False
for i, line in enumerate(src):
    out.append(line)
    if re.match(r"_ZN\S*k_near_tie_runs\S*:", line):
        inside = True
    if inside and line.strip() == "s_endpgm":
        inside = False
    if inside and not done and line.strip() == "global_store_dword v[0:1], v4, off":
        rec = ["s_add_u32 s2, s2, 0x1000", "s_addc_u32 s3, s3, 0",
               "v_lshlrev_b32_e32 v26, 7, v12", "v_mov_b32_e32 v27, 0", "s_nop 1",
               "v_lshl_add_u64 v[0:1], v[26:27], 0, s[2:3]", "s_mov_b64 s[20:21], exec"]
        words = ["s18", "s19", "s10", "s11", "s4", "s5", "s6", "s7", "s8", "s9", "v2", "v4", "v16", "s20", "s21"]
        for j, w in enumerate(words):
            if w.startswith("s"):
                rec += ["v_mov_b32_e32 v20, %s" % w, "s_nop 1", "global_store_dword v[0:1], v20, off offset:%d" % (4 * j), "s_nop 1"]
            else:
                rec += ["global_store_dword v[0:1], %s, off offset:%d" % (w, 4 * j)]
        rec += ["v_or_b32_e32 v20, 0xabcd0000, v12", "s_nop 1", "global_store_dword v[0:1], v20, off offset:60", "s_nop 1"]
        # far[lo - 1] again (lo = v4; lanes with lo == 0 read far[0]); VGPR tuples are 64-bit aligned on gfx950
        rec += ["v_max_u32_e32 v20, 1, v4", "v_add_u32_e32 v20, -1, v20", "v_mov_b32_e32 v21, 0", "s_nop 1",
                "v_lshl_add_u64 v[22:23], v[20:21], 2, s[8:9]",
                "global_load_dword v24, v[22:23], off", "global_load_dword v25, v[22:23], off sc0 sc1", "s_waitcnt vmcnt(0)",
                "global_store_dword v[0:1], v24, off offset:64", "global_store_dword v[0:1], v25, off offset:68",
                # pk word of that suffix: pk + (f >> 5) * 8
                "v_lshrrev_b32_e32 v20, 5, v24", "v_mov_b32_e32 v21, 0", "s_nop 1",
                "v_lshl_add_u64 v[22:23], v[20:21], 3, s[10:11]",
                "global_load_dwordx2 v[28:29], v[22:23], off", "s_waitcnt vmcnt(0)",
                "global_store_dwordx2 v[0:1], v[28:29], off offset:72", "s_nop 1",
                "global_load_dwordx2 v[28:29], v[22:23], off sc0 sc1", "s_waitcnt vmcnt(0)",
                "global_store_dwordx2 v[0:1], v[28:29], off offset:80", "s_nop 1",
                # pk word of pe (v2)
                "v_lshrrev_b32_e32 v20, 5, v2", "v_mov_b32_e32 v21, 0", "s_nop 1",
                "v_lshl_add_u64 v[22:23], v[20:21], 3, s[10:11]",
                "global_load_dwordx2 v[28:29], v[22:23], off", "s_waitcnt vmcnt(0)",
                "global_store_dwordx2 v[0:1], v[28:29], off offset:88", "s_nop 1",
                "global_load_dwordx2 v[28:29], v[22:23], off sc0 sc1", "s_waitcnt vmcnt(0)",
                "global_store_dwordx2 v[0:1], v[28:29], off offset:96"]
        out += ["\t" + r for r in rec]
        done = True
assert done, "k_near_tie_runs: final store not found"
open(sys.argv[2], "w").write("\n".join(out))
print("patched")
