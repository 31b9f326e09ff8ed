"""First-contact GPU script: runs a few sizes and prints stage-by-stage diagnostics."""
import sys, time, traceback
import numpy as np
sys.path.insert(0, '.')
import kiss_amd
from tests import gen, oracle_binding
orc = oracle_binding.load()
ctx = kiss_amd.Context(max_n=6_000_000)
print("workspace MB", ctx.workspace_bytes() / 1e6, flush=True)
def run(name, S, k):
    S = np.ascontiguousarray(S, np.uint8); n = S.size
    t = time.time()
    try:
        sa = ctx.suffix_sort(S, k)
    except Exception as e:
        print(name, n, k, "EXC", e, flush=True); return False
    dt = time.time() - t
    ref, lref = orc.suffix_sort(S, k, stages=True)
    st = ctx.stats()
    ok = True
    if n:
        asc, srt, counts = ctx.stage_outputs()
        lr, hist = orc.get_lms(S)
        a_ok = asc.size == lr.size - 1 and np.array_equal(asc, lr[:-1])
        c_ok = np.array_equal(counts[0:4], hist[4,:4]) and np.array_equal(counts[8:12], hist[2,:4])
        s_ok = srt.size == lref.size - 1 and np.array_equal(srt, lref[1:])
        f_ok = np.array_equal(sa, ref)
        ok = a_ok and c_ok and s_ok and f_ok
        print("%-22s n=%8d k=%10d lms_asc=%s counts=%s lms_sorted=%s SA=%s  m=%d rounds=%d passes=%d near=%d  %.1f ms (dev %.2f: pack %.2f cls %.2f sort %.2f place %.2f ind %.2f)" % (
            name, n, k, a_ok, c_ok, s_ok, f_ok, st['m'], st['lms_rounds'], st['induce_passes'], st['near_end'], dt*1e3,
            st['ms_total'], st['ms_pack'], st['ms_classify'], st['ms_lms_sort'], st['ms_place'], st['ms_induce']), flush=True)
        if not a_ok:
            print("   asc size", asc.size, lr.size - 1, "first diff", (np.nonzero(asc[:min(asc.size, lr.size-1)] != lr[:min(asc.size, lr.size-1)])[0][:5]), flush=True)
            print("   counts", counts, hist[4,:4], hist[2,:4])
        elif not s_ok:
            bad = np.nonzero(srt != lref[1:])[0]; print("   sorted lms bad", bad.size, bad[:5], srt[bad[:5]], lref[1:][bad[:5]], flush=True)
        elif not f_ok:
            bad = np.nonzero(sa != ref)[0]; print("   SA bad", bad.size, bad[:8], sa[bad[:8]], ref[bad[:8]], flush=True)
    else:
        ok = np.array_equal(sa, ref); print(name, n, k, "SA", ok)
    return ok
allok = True
for n in [0, 1, 2, 5, 33, 200, 1000, 5000, 100_003, 1_000_000]:
    for k in [256, 32, 0xFFFFFFFF]:
        allok &= run("iid", gen.iid(n, 1000 + n), k)
for per in [1, 2, 3, 7, 37, 400]:
    for k in [256, 0xFFFFFFFF]:
        allok &= run("periodic%d" % per, gen.periodic(20000, per, 7 + per, 6), k)
allok &= run("allA", np.zeros(3000, np.uint8), 256)
allok &= run("allT", np.full(3000, 3, np.uint8), 256)
allok &= run("genome2M", gen.genome_like(2_000_000, 11), 256)
allok &= run("genome2M", gen.genome_like(2_000_000, 11), 32)
allok &= run("iid5M", gen.iid(5_000_000, 5), 256)
print("ALL OK" if allok else "FAILURES", flush=True)
