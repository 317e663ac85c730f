"""Static hazard check of a gfx950 ISA listing (hipcc -S output) for the kernels that issue LDS / SMEM
loads -- and, since round 2, vector-memory loads -- from inline asm.

The compiler does not know that the destination registers of an inline-asm `ds_read*` / `s_load*`
are still in flight: it may copy them, or hand a dead part of them out as a temporary, before the
hand-placed `s_waitcnt`.  This script replays every basic block of a kernel, tracks which registers
are in flight (exactly inside a block -- LDS returns in order, so `lgkmcnt(n)` retires all but the
youngest n; conservatively across blocks -- only `lgkmcnt(0)` retires what a predecessor left in
flight) and reports every instruction that reads or writes such a register.

    python tools/check_inflight.py file.s [kernel-name-regex]
"""
import re
import sys

_V = re.compile(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b')
_S = re.compile(r'\bs\[(\d+):(\d+)\]|\bs(\d+)\b')
_BR = re.compile(r'^(s_cbranch_\w+|s_branch)\s+(\S+)')


def _regs(rx, tok):
    out = set()
    for m in rx.finditer(tok):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def split_kernels(text):
    """{name: [instruction lines]} for every function of the listing."""
    kernels, name, body = {}, None, []
    for line in text.splitlines():
        m = re.match(r'^(_Z\w+|\w+):\s*(;.*)?$', line)
        if m and not line.startswith('.'):
            if name is not None:
                kernels[name] = body
            name, body = m.group(1), []
            continue
        if name is None:
            continue
        t = line.split(';')[0].strip()
        if not t:
            if '#ASMSTART' in line or '#ASMEND' in line:     # inline-asm brackets: kept as markers
                body.append('#ASMSTART' if '#ASMSTART' in line else '#ASMEND')
            continue
        if t.startswith('.') and not t.endswith(':'):
            if t.startswith('.Lfunc_end'):
                kernels[name] = body
                name, body = None, []
            continue
        body.append(t)   # a kernel may hold several s_endpgm (early exits): .Lfunc_end closes it
    if name is not None:
        kernels[name] = body
    return kernels


def _blocks(lines):
    blocks, cur, label = [], [], '<entry>'
    for t in lines:
        if t.startswith('#ASM'):
            cur.append(t)
            continue
        if t.endswith(':'):
            blocks.append((label, cur))
            label, cur = t[:-1], []
            continue
        cur.append(t)
        if _BR.match(t) or t.startswith('s_endpgm') or t.startswith('s_setpc'):
            blocks.append((label, cur))
            label, cur = None, []
    blocks.append((label, cur))
    blocks = [(l, b) for l, b in blocks if b or l]
    index = {l: i for i, (l, _) in enumerate(blocks) if l}
    succ = []
    for i, (l, b) in enumerate(blocks):
        s = set()
        last = b[-1] if b else ''
        m = _BR.match(last)
        if m and m.group(2) in index:
            s.add(index[m.group(2)])
        if not (last.startswith('s_branch') or last.startswith('s_endpgm') or last.startswith('s_setpc')):
            if i + 1 < len(blocks):
                s.add(i + 1)
        succ.append(s)
    return blocks, succ


def _run_block(lines, state, report=None):
    """Replay one block from `state` = (queue, v_un, s_un): `queue` is the in-order list of ops in
    flight that every predecessor agrees on, v_un / s_un are registers some predecessor may have
    left in flight in an unknown order (only lgkmcnt(0) retires those).  Returns the end state."""
    q = [(k, set(d)) for k, d in state[0]]
    v_un, s_un = set(state[1]), set(state[2])
    vq = [set(d) for d in state[3]] if len(state) > 3 else []     # vector-memory ops, in order (vmcnt)
    vm_un = set(state[4]) if len(state) > 4 else set()
    in_asm = False
    for t in lines:
        if t.startswith('#ASM'):
            in_asm = t == '#ASMSTART'
            continue
        op = t.split()[0]
        args = t[len(op):]
        if op == 's_waitcnt':
            m = re.search(r'lgkmcnt\((\d+)\)', t)
            if m:
                n = int(m.group(1))
                if n == 0:
                    q, v_un, s_un = [], set(), set()
                elif not v_un:
                    # at most n operations are outstanding in total.  LDS operations return in
                    # order among themselves, scalar loads in any order: whatever the scalar loads
                    # do, at most the n youngest LDS operations can still be in flight; every
                    # scalar load may be (only lgkmcnt(0) proves one complete)
                    n_lds = sum(1 for k, _ in q if k == 'lds')
                    drop = max(0, n_lds - n)
                    kept = []
                    for k, d in q:
                        if k == 'lds' and drop > 0:
                            drop -= 1
                            continue
                        kept.append((k, d))
                    q = kept
            m = re.search(r'vmcnt\((\d+)\)', t)
            if m:
                n = int(m.group(1))
                if n == 0:
                    vq, vm_un = [], set()
                elif not vm_un:
                    while len(vq) > n:     # loads and stores return in order on gfx9
                        vq.pop(0)
            continue
        v_fl = set(v_un).union(*[d for k, d in q if k == 'lds']).union(vm_un, *vq)
        s_fl = set(s_un).union(*[d for k, d in q if k == 'smem'])
        v_t, s_t = _regs(_V, args), _regs(_S, args)
        hit = (v_t & v_fl, s_t & s_fl)
        if (hit[0] or hit[1]) and report is not None:
            report(t, sorted(hit[0]), sorted(hit[1]))
        if op.startswith('ds_read') or op.startswith('ds_load'):
            q.append(('lds', _regs(_V, args.split(',')[0])))
        elif op.startswith('ds_'):
            q.append(('lds', set()))
        elif op.startswith('s_load') or op.startswith('s_buffer_load'):
            q.append(('smem', _regs(_S, args.split(',')[0])))
        elif re.match(r'(global|flat|buffer|scratch)_load', op) and ' lds' not in t:
            # destinations are tracked for hand-issued (inline-asm) loads only: the compiler counts
            # its own loads itself; they still take a slot of the in-order queue
            vq.append(_regs(_V, args.split(',')[0]) if in_asm else set())
        elif re.match(r'(global|flat|buffer|scratch)_(store|atomic)', op) or (' lds' in t and 'load' in op):
            vq.append(set())               # counts in vmcnt, has no destination registers
    # normal form of the vector-memory queue: at most 64 entries (vmcnt is a 6-bit counter: an
    # older operation has completed), and no leading entries without tracked registers (dropping
    # them only makes later vmcnt(n) retire less)
    vq = vq[-64:]
    while vq and not vq[0]:
        vq.pop(0)
    q = q[-16:]                    # lgkmcnt is a 4-bit counter
    return (tuple((k, frozenset(d)) for k, d in q), frozenset(v_un), frozenset(s_un),
            tuple(frozenset(d) for d in vq), frozenset(vm_un))


def _merge_queue(qa, qb, kinded):
    """Element-wise merge of two in-order queues, aligned at their YOUNG end: slot i from the end
    retires at the same counter value in both predecessors, so the union of the two destination
    sets is a sound model of it; older entries that only one predecessor has are kept (they can
    only make a later wait retire less).  Returns None if the kinds (LDS / SMEM) disagree."""
    if len(qa) < len(qb):
        qa, qb = qb, qa
    out = list(qa)
    off = len(qa) - len(qb)
    for i, e in enumerate(qb):
        if kinded:
            if out[off + i][0] != e[0]:
                return None
            out[off + i] = (e[0], out[off + i][1] | e[1])
        else:
            out[off + i] = out[off + i] | e
    return tuple(out)


def _merge(a, b):
    if a is None:
        return b
    vq = _merge_queue(a[3], b[3], False)
    vm = a[4] | b[4]
    q = _merge_queue(a[0], b[0], True)
    if q is not None:
        return (q, a[1] | b[1], a[2] | b[2], vq, vm)
    v = set(a[1] | b[1]).union(*[d for k, d in a[0] + b[0] if k == 'lds'])
    s = set(a[2] | b[2]).union(*[d for k, d in a[0] + b[0] if k == 'smem'])
    return ((), frozenset(v), frozenset(s), vq, vm)


def check_kernel(lines):
    """List of (block, instruction, vgprs, sgprs) violations of one kernel."""
    blocks, succ = _blocks(lines)
    n = len(blocks)
    entry = [None] * n
    entry[0] = ((), frozenset(), frozenset(), (), frozenset())
    work = [0]
    while work:
        i = work.pop(0)
        out = _run_block(blocks[i][1], entry[i])
        for j in succ[i]:
            merged = _merge(entry[j], out)
            if merged != entry[j]:
                entry[j] = merged
                if j not in work:
                    work.append(j)
    found = []
    for i in range(n):
        if entry[i] is None:
            continue
        _run_block(blocks[i][1], entry[i],
                   report=lambda t, v, s, lab=blocks[i][0]: found.append((lab, t, v, s)))
    return found


def main(argv):
    text = open(argv[1]).read()
    pat = re.compile(argv[2]) if len(argv) > 2 else None
    total = 0
    for name, lines in split_kernels(text).items():
        if pat and not pat.search(name):
            continue
        if not any('ds_read' in t or 's_load' in t or 'global_load' in t for t in lines):
            continue
        bad = check_kernel(lines)
        total += len(bad)
        for lab, t, v, s in bad[:8]:
            print(f"{name[:60]} [{lab}]: {t}   <-- in flight: v{v} s{s}")
    print("violations:", total)
    return 1 if total else 0


if __name__ == '__main__':
    sys.exit(main(sys.argv))
