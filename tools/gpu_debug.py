"""GPU diagnostics: parity + per-kernel-class timing for selected cases."""
import sys, time
import numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kiss_amd
from tests import gen, oracle_binding
orc = oracle_binding.load()
MAXN = int(sys.argv[1]) if len(sys.argv) > 1 else 6_000_000
ctx = kiss_amd.Context(max_n=MAXN)
print("workspace MB", ctx.workspace_bytes() / 1e6, flush=True)
def run(name, S, k, prof=False, check=True, reps=1):
    S = np.ascontiguousarray(S, np.uint8); n = S.size
    ctx.set_profiling(prof)
    for _ in range(reps):
        t = time.time(); sa = ctx.suffix_sort(S, k); dt = time.time() - t
    st = ctx.stats()
    ok = None
    if check:
        ref = orc.suffix_sort(S, k); ok = bool(np.array_equal(sa, ref))
    print("%-14s n=%9d k=%10d SA=%s m=%d rounds=%d itemrounds=%d passes=%d near=%d wall %.1f ms dev %.2f: pack %.2f cls %.2f sort %.2f place %.2f ind %.2f  -> %.1f Mbases/s" % (
        name, n, k, ok, st['m'], st['lms_rounds'], st['sort_item_rounds'], st['induce_passes'], st['near_end'], dt*1e3,
        st['ms_total'], st['ms_pack'], st['ms_classify'], st['ms_lms_sort'], st['ms_place'], st['ms_induce'], n / max(st['ms_total'],1e-9) / 1e3), flush=True)
    if prof:
        for kname, v in st['kernels'].items():
            if v['launches']:
                print("      %-15s ms %9.3f launches %6d items %12d  avg %.1f us" % (kname, v['ms'], v['launches'], v['items'], 1e3*v['ms']/v['launches']), flush=True)
    return ok
run("warm", gen.iid(100000, 1), 256)
run("periodic1", gen.periodic(20000, 1, 8, 6), 256, prof=True)
run("periodic1", gen.periodic(20000, 1, 8, 6), 256, prof=False)
run("periodic7", gen.periodic(20000, 7, 14, 6), 256, prof=True)
run("periodic400", gen.periodic(20000, 400, 407, 6), 256, prof=True)
run("periodic400", gen.periodic(20000, 400, 407, 6), 256, prof=False)
G = gen.genome_like(2_000_000, 11)
run("genome2M", G, 256, prof=True)
run("genome2M", G, 256, prof=False)
if MAXN >= 50_000_000:
    S = gen.iid(50_000_000, 5)
    run("iid50M", S, 256, reps=2)
    run("iid50M", S, 256, prof=True, check=False)
    G = gen.genome_like(50_000_000, 1)
    run("genome50M", G, 256, reps=2)
    run("genome50M", G, 256, prof=True, check=False)
    run("genome50M", G, 0xFFFFFFFF, check=True)
