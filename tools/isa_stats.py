"""Instruction mix per kernel from a hipcc -S listing (development aid)."""
import re, collections, sys
txt = open(sys.argv[1]).read().split('\n')
cur = None; stats = {}
for l in txt:
    m = re.match(r'^(_ZN3rtd\w+):', l)
    if m: cur = m.group(1); stats[cur] = collections.Counter(); continue
    if l.startswith('.Lfunc_end'): cur = None
    if cur and l.startswith('\t') and not l.strip().startswith(('.', ';')):
        stats[cur][l.strip().split()[0]] += 1
for name, c in stats.items():
    tot = sum(c.values())
    g = lambda p: sum(v for k, v in c.items() if k.startswith(p))
    print(f"{name[8:48]:40s} total {tot:5d} valu {g('v_'):5d} div_scale {c['v_div_scale_f32']:3d} rcp {c['v_rcp_f32']:3d} sqrt {c['v_sqrt_f32']:3d} rsq {c['v_rsq_f32']:3d} "
          f"atomic {sum(v for k, v in c.items() if 'atomic' in k):2d} cbranch {g('s_cbranch'):3d} gload {g('global_load'):3d} gstore {g('global_store'):3d} sload {g('s_load'):3d} scratch {g('scratch_'):3d}")
