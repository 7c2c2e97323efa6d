"""ISA helper: python tools/dbg/isa_loop.py <file.s> <mangled-name-fragment> -> main loop extent, scratch ops, opcode histogram."""
import collections, re, sys
s = open(sys.argv[1]).read()
frag = sys.argv[2]
name = [m for m in re.findall(r'^(_Z\w+):', s, re.M) if frag in m][0]
i = s.index(name + ':'); j = s.index('.Lfunc_end', i)
body = s[i:j].splitlines()
labels = {}
for n, l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[m.group(1)] = n
loops = []
for n, l in enumerate(body):
    m = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)', l)
    if m:
        t = m.group(1) or m.group(2)
        if t in labels and labels[t] < n: loops.append((labels[t], n))
main = max(loops, key=lambda x: x[1] - x[0])
print('lines', len(body), 'main loop', main)
for n, l in enumerate(body):
    if 'scratch' in l: print(n, 'IN-LOOP' if main[0] <= n <= main[1] else '       ', l.strip()[:80])
c = collections.Counter()
for l in body[main[0]:main[1]]:
    l = l.strip()
    if not l or l.startswith(('.', ';')) or l.endswith(':'): continue
    c[l.split()[0]] += 1
print('loop instrs', sum(c.values()), {k: c[k] for k in ('v_readlane_b32', 'v_writelane_b32', 's_nop', 's_and_saveexec_b64', 's_waitcnt', 'ds_read_b32', 'ds_write_b32', 'global_store_dword')})
