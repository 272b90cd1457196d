"""CPU model of prefill64_kernel's per-wave control flow (mfa_prefill64.hip main loop): for a list of work items it replays
which block every wave runs in every iteration and checks what the hardware cannot forgive -- all four waves execute the
same number of workgroup barriers, every tile's K / V pieces are requested exactly once by every wave before the
iteration that reads them, no ring slot is overwritten while a wave may still read it, every tile of every wave gets
exactly one scores / softmax / P.V phase in that order, and every item's epilogue runs once.  Used by
tests/test_p64_model_cpu.py; keep it in step with the kernel."""
import itertools

BIG = 0x3fffffff


def ctx(m0, wave, sq, sk, has_hi, hi):
    wrow0 = m0 + 64 * wave
    last_wg = min(m0 + 256, sq) - 1
    nt = (min(sk - 1, last_wg + hi) if has_hi else sk - 1) // 64 + 1
    last_w = min(wrow0 + 63, sq - 1)
    nt_w = 0 if wrow0 >= sq else (min(sk - 1, last_w + hi) if has_hi else sk - 1) // 64 + 1
    jm = sk // 64 if sk % 64 else BIG
    if has_hi:
        jm = min(jm, (wrow0 + hi + 1) // 64)
    return dict(nt=nt, nt_w=nt_w, nm=max(nt_w - jm, 0), wrow0=wrow0)


def run_wave(items, wave, sq, sk, has_hi, hi, no_loop=False):
    """items: list of m0 (the (batch, head) does not matter for control flow).  Returns the event list of the wave."""
    ev = []  # (kind, ...) in program order
    G0, warm, pend = 0, False, False
    for idx, m0 in enumerate(items):
        cc = ctx(m0, wave, sq, sk, has_hi, hi)
        has_next = idx + 1 < len(items)
        cn = ctx(items[idx + 1], wave, sq, sk, has_hi, hi) if has_next else cc
        warm_next = has_next and cc["nt"] >= 3 and cn["nt"] >= 3
        nt, nt_w, nm = cc["nt"], cc["nt_w"], cc["nm"]
        gl = G0 + nt - 1
        ga = G0 + nt - nt_w if nt_w > 0 else BIG
        mine = lambda g: ga <= g <= gl
        tile_of = lambda g: nt - 1 - (g - G0)
        q_is_next = False
        if not warm:
            ev.append(("barrier",))
            ev.append(("dmaK", idx, nt - 1, G0))
            if nt >= 2:
                ev.append(("dmaK", idx, nt - 2, G0 + 1))
            ev.append(("dmaV", idx, nt - 1, G0))
            ev.append(("dmaQ", idx)); ev.append(("qswap", idx))
            ev.append(("barrier",))
            if mine(G0):
                ev.append(("X", idx, G0))
        g = G0
        while g <= gl:
            has_y = mine(g - 1) if g > G0 else (warm and pend)
            sm = "none" if not mine(g) else "first" if g == ga else "mnormal" if g < ga + nm else "normal"
            has_x = mine(g + 1) if g < gl else (warm_next and cn["nt_w"] > 0 and cn["nt_w"] == cn["nt"])
            k_cur, v_cur = g + 2 <= gl, g + 1 <= gl
            k_any, v_any = k_cur or warm_next, v_cur or warm_next
            k_tile = tile_of(g + 2) if k_cur else cn["nt"] - 1 - (g + 2 - (gl + 1))
            v_tile = tile_of(g + 1) if v_cur else cn["nt"] - 1 - (g + 1 - (gl + 1))
            if g == gl and warm_next and cn["nt_w"] > 0 and not q_is_next:
                ev.append(("qswap", idx + 1)); q_is_next = True
            first_a_done = g == ga and g > G0  # (the wave's first tile: its scores, mask and row maximum came in iteration g - 1)
            if (sm == "normal" or (sm == "first" and first_a_done)) and has_x and not no_loop:
                jend = gl - 1 if g <= gl - 2 else g + 1
                for j in range(g, jend):
                    ev.append(("barrier",))
                    kt = k_tile - (j - g); vt = v_tile - (j - g)
                    if k_any: ev.append(("dmaK", idx if k_cur else idx + 1, kt, j + 2))
                    else: ev.append(("dmaK0", j + 2))
                    if v_any: ev.append(("dmaV", idx if v_cur else idx + 1, vt, j + 1))
                    else: ev.append(("dmaV0", j + 1))
                    if j - 1 >= ga: ev.append(("Y", idx, j - 1))  # (else: the empty tile before the wave's first)
                    ev.append(("SM", idx, j)); ev.append(("X", idx if j + 1 <= gl else idx + 1, j + 1))
                g = jend
                continue
            ev.append(("barrier",))
            if k_any: ev.append(("dmaK", idx if k_cur else idx + 1, k_tile, g + 2))
            if v_any: ev.append(("dmaV", idx if v_cur else idx + 1, v_tile, g + 1))
            if sm in ("normal", "mnormal"):
                ev.append(("Y", idx, g - 1)); ev.append(("SM", idx, g))
                if has_x: ev.append(("X", idx if g + 1 <= gl else idx + 1, g + 1))
            elif sm == "first" and has_y and has_x:
                ev.append(("Y", idx - 1, g - 1)); ev.append(("SM", idx, g)); ev.append(("X", idx, g + 1)); ev.append(("epi", idx - 1))
            else:
                if has_y:
                    ev.append(("Y", idx - 1, g - 1)); ev.append(("epi", idx - 1))
                if sm == "first":
                    ev.append(("SM", idx, g))
                    if has_x: ev.append(("X", idx if g + 1 <= gl else idx + 1, g + 1))
                elif has_x:
                    ev.append(("X", idx if g + 1 <= gl else idx + 1, g + 1))
            if g == G0 and warm_next and cn["nt_w"] > 0:
                ev.append(("dmaQ", idx + 1))
            g += 1
        if not warm_next:
            if nt_w > 0:
                ev.append(("Y", idx, gl)); ev.append(("epi", idx))
            pend = False
        else:
            pend = nt_w > 0
        warm = warm_next
        G0 = gl + 1
    return ev


def check(items, sq, sk, has_hi, hi, no_loop=False):
    waves = [run_wave(items, w, sq, sk, has_hi, hi, no_loop) for w in range(4)]
    # 1. barriers and DMA requests: identical sequences in every wave (each wave moves a quarter of every tile)
    sync = [[e for e in ev if e[0] in ("barrier", "dmaK", "dmaV")] for ev in waves]  # (dmaK0 / dmaV0: the loop block's requests through the null descriptor move nothing)
    assert all(s == sync[0] for s in sync), "waves disagree on barriers / tile requests"
    # 2. per wave: every tile it owns gets X, SM, Y once, in order; one epilogue per item with rows; Q in registers is the item's
    for w, ev in enumerate(waves):
        G0 = 0
        for idx, m0 in enumerate(items):
            c = ctx(m0, w, sq, sk, has_hi, hi)
            for g in range(G0 + c["nt"] - c["nt_w"], G0 + c["nt"]):
                pos = [next(i for i, e in enumerate(ev) if e == (k, idx, g)) for k in ("X", "SM", "Y")]
                assert pos == sorted(pos) and all(ev.count((k, idx, g)) == 1 for k in ("X", "SM", "Y")), (w, idx, g, pos)
                # Q: the last swap before X must be this item's
                qs = [e[1] for e in ev[:pos[0]] if e[0] == "qswap"]
                assert qs and qs[-1] == idx, (w, idx, g, qs[-3:])
            assert ev.count(("epi", idx)) == (1 if c["nt_w"] > 0 else 0), (w, idx)
            G0 += c["nt"]
        assert sum(1 for e in ev if e[0] in ("X", "SM", "Y")) == 3 * sum(ctx(m0, w, sq, sk, has_hi, hi)["nt_w"] for m0 in items)
    # 3. ring hazards, on the common request sequence: tile number n's K must be requested after the barrier of iteration n-1
    # ... (slot of K(n-3), read in iteration n-4's X and by a redo after iteration n-3) and have a barrier between request and
    # the X that reads it; V(n) after the barrier of iteration n-1 (slot of V(n-3), read by Y in iteration n-2)
    for w, ev in enumerate(waves):
        nbar = 0
        req_k, req_v = {}, {}
        for e in ev:
            if e[0] == "barrier": nbar += 1
            elif e[0] == "dmaK": assert e[3] not in req_k; req_k[e[3]] = nbar
            elif e[0] == "dmaV": assert e[3] not in req_v; req_v[e[3]] = nbar
            elif e[0] == "X": assert e[2] in req_k and req_k[e[2]] < nbar, ("K not landed", w, e)
            elif e[0] == "Y": assert e[2] in req_v and req_v[e[2]] < nbar, ("V not landed", w, e)
    # every wave's reads of a slot precede (by a barrier) the request that overwrites it: compare across waves by barrier index
    last_read_k, last_read_v = {}, {}
    for ev in waves:
        nbar = 0
        for e in ev:
            if e[0] == "barrier": nbar += 1
            elif e[0] in ("X", "SM"): last_read_k[e[2]] = max(last_read_k.get(e[2], 0), nbar)  # (SM: a redo of tile g reads K(g) again)
            elif e[0] == "Y": last_read_v[e[2]] = max(last_read_v.get(e[2], 0), nbar)
    nbar = 0
    for e in waves[0]:
        if e[0] == "barrier": nbar += 1
        elif e[0] == "dmaK" and e[3] - 3 in last_read_k: assert last_read_k[e[3] - 3] < nbar, ("K slot overwritten early", e)
        elif e[0] == "dmaV" and e[3] - 3 in last_read_v: assert last_read_v[e[3] - 3] < nbar, ("V slot overwritten early", e)
    return len(sync[0])


if __name__ == "__main__":
    n = 0
    for sq, sk, causal in [(1024, 1024, True), (1024, 1024, False), (256, 256, True), (200, 200, True), (300, 333, False), (1, 77, False),
                           (1000, 1000, True), (257, 511, False), (129, 64, False), (512, 512, True), (64, 64, True), (2048, 2048, True),
                           (320, 1000, True), (700, 710, True)]:
        has_hi, hi = causal, 0
        nmb = (sq + 255) // 256
        blocks = [m * 256 for m in range(nmb)]
        for items in (blocks[::-1], blocks, blocks[::-1] + blocks, blocks * 3, [blocks[0]], [blocks[-1]] * 2):
            for nl in (False, True):
                n += check(items, sq, sk, has_hi, hi, nl)
        # bottom-right alignment (kv-cache style, sk >= sq)
        if sk >= sq:
            check(blocks[::-1] * 2, sq, sk, True, sk - sq)
    print("ok", n)
