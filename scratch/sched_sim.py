import sys, pickle, heapq
sys.path.insert(0, '/root/repo')
import numpy as np
exec(open('scratch/len_feat.py').read().split("for name, f in")[0])   # features
trips = pickle.load(open('scratch/trips.pkl', 'rb'))
OV = 1.5   # per-trip overhead in KKT units (linearize, checks)
cost = [t[0].astype(float) + OV for t in trips]
omega = [t[1] for t in trips]
tot = np.array([c.sum() for c in cost])
print('total', tot.sum(), 'balanced', tot.sum() / 1024, 'longest', tot.max())
def level(b, ntrips_done):
    om = omega[b][ntrips_done]   # omega after that many trips (hist index)
    lvl = 0; x = 1.5
    while x < om and lvl < 15: lvl += 1; x *= 5.0
    return lvl
def sim(order, probe, unsliced_first=0, S=1024):
    # event simulation: slots become free at times; fresh list (ordered), level lists
    fresh = list(order); fi = 0
    lists = [[] for _ in range(16)]
    probing = 0
    t_free = [(0.0, s) for s in range(S)]
    heapq.heapify(t_free)
    pending = []   # (time, b, done_trips, visits) pushes that become visible at time
    done_t = 0.0
    prob_left = len(fresh)   # problems that may still be pushed
    idle = []
    import itertools
    cnt = itertools.count()
    while t_free:
        t, s = heapq.heappop(t_free)
        # make visible pushes up to t
        while pending and pending[0][0] <= t:
            _, _, b, d, v = heapq.heappop(pending); lists[level(b, d)].append((b, d, v))
        job = None
        if fi < len(fresh):
            b = fresh[fi]; fi += 1; job = (b, 0, 0, fi <= unsliced_first)
        else:
            for L in range(15, -1, -1):
                if lists[L]: b, d, v = lists[L].pop(0); job = (b, d, v, False); break
        if job is None:
            if pending:   # wait for next push
                heapq.heappush(t_free, (pending[0][0], s)); continue
            continue   # retire
        b, d, v, uns = job
        nt = len(cost[b])
        if not uns and v < probe and probe > 0:
            run = 1
        else:
            run = nt - d
        dur = cost[b][d:d + run].sum()
        te = t + dur
        d2 = d + run
        if d2 < nt:
            heapq.heappush(pending, (te, next(cnt), b, d2, v + 1))
        done_t = max(done_t, te)
        heapq.heappush(t_free, (te, s))
    return done_t
ids = np.arange(B)
print('FCFS', sim(ids, 0))
print('current (probe 2)', sim(ids, 2), 'probe 1', sim(ids, 1))
o_max = np.argsort(-f_max, kind='stable')
print('static max unsliced', sim(o_max, 0))
for q in (256, 512, 768, 1024, 1536):
    print('hybrid q', q, 'probe2', sim(o_max, 2, q), 'probe1', sim(o_max, 1, q))
print('perfect LPT', sim(np.argsort(-tot), 0))

def sim2(order, prio, S=1024, fresh_rank=0.5):
    """always slice per trip; prio(b, done_trips) -> larger = sooner; fresh problems have priority fresh_rank"""
    import itertools
    cnt = itertools.count()
    fresh = list(order); fi = 0
    ready = []    # heap of (-prio, seq, b, d)
    pending = []  # (time, seq, b, d)
    t_free = [(0.0, s) for s in range(S)]
    heapq.heapify(t_free)
    done_t = 0.0
    while t_free:
        t, s = heapq.heappop(t_free)
        while pending and pending[0][0] <= t:
            _, _, b, d = heapq.heappop(pending); heapq.heappush(ready, (-prio(b, d), next(cnt), b, d))
        job = None
        if ready and (-ready[0][0] > fresh_rank or fi >= len(fresh)):
            _, _, b, d = heapq.heappop(ready); job = (b, d)
        elif fi < len(fresh):
            job = (fresh[fi], 0); fi += 1
        if job is None:
            if pending: heapq.heappush(t_free, (pending[0][0], s))
            continue
        b, d = job
        te = t + cost[b][d]
        if d + 1 < len(cost[b]): heapq.heappush(pending, (te, next(cnt), b, d + 1))
        done_t = max(done_t, te)
        heapq.heappush(t_free, (te, s))
    return done_t
print('--- always slice')
print('V1 level>=1 before fresh, level0 after', sim2(ids, lambda b, d: level(b, d)))
print('V1 + order by f_max', sim2(o_max, lambda b, d: level(b, d)))
print('V1b level then trips done', sim2(ids, lambda b, d: level(b, d) + 0.01 * d))
print('V1c level*, level0 by fewer trips done', sim2(ids, lambda b, d: level(b, d) - 0.01 * d))
print('V3 all lists after fresh', sim2(ids, lambda b, d: 0.1 * level(b, d) / 16))
print('oracle remaining work', sim2(ids, lambda b, d: 1 + cost[b][d:].sum()))
print('--- diagnostics')
# perfect knowledge preemptive: fresh sorted by total desc, ready prio = remaining, fresh taken only if its total > best ready remaining
def sim3(S=1024):
    import itertools
    cnt = itertools.count()
    ready = [(-tot[b], next(cnt), b, 0) for b in range(B)]
    heapq.heapify(ready)
    pending = []
    t_free = [(0.0, s) for s in range(S)]
    done_t = 0
    while t_free:
        t, s = heapq.heappop(t_free)
        while pending and pending[0][0] <= t:
            _, _, b, d = heapq.heappop(pending); heapq.heappush(ready, (-cost[b][d:].sum(), next(cnt), b, d))
        if not ready:
            if pending: heapq.heappush(t_free, (pending[0][0], s))
            continue
        _, _, b, d = heapq.heappop(ready)
        te = t + cost[b][d]
        if d + 1 < len(cost[b]): heapq.heappush(pending, (te, next(cnt), b, d + 1))
        done_t = max(done_t, te)
        heapq.heappush(t_free, (te, s))
    return done_t
print('perfect preemptive', sim3())
long_ = np.argsort(-tot)[:40]
for b in long_[:40]:
    om = omega[b]
    first_raise = next((i for i in range(len(om)) if om[i] > 1.5), -1)
    print(b, 'tot', tot[b], 'trips', len(cost[b]), 'first omega raise at trip', first_raise, 'max lvl', level(b, len(om) - 1), 'fmax rank', int(np.where(o_max == b)[0][0]))
print('--- trace V1+fmax')
def sim2t(order, prio, S=1024, fresh_rank=0.5):
    import itertools
    cnt = itertools.count()
    fresh = list(order); fi = 0
    ready = []; pending = []
    t_free = [(0.0, s) for s in range(S)]
    heapq.heapify(t_free)
    start = {}; end = {}; waits = np.zeros(B)
    lastend = {}
    while t_free:
        t, s = heapq.heappop(t_free)
        while pending and pending[0][0] <= t:
            _, _, b, d = heapq.heappop(pending); heapq.heappush(ready, (-prio(b, d), next(cnt), b, d))
        job = None
        if ready and (-ready[0][0] > fresh_rank or fi >= len(fresh)):
            _, _, b, d = heapq.heappop(ready); job = (b, d)
        elif fi < len(fresh):
            job = (fresh[fi], 0); fi += 1
        if job is None:
            if pending: heapq.heappush(t_free, (pending[0][0], s))
            continue
        b, d = job
        if d == 0: start[b] = t
        else: waits[b] += t - lastend[b]
        te = t + cost[b][d]
        lastend[b] = te
        if d + 1 < len(cost[b]): heapq.heappush(pending, (te, next(cnt), b, d + 1))
        else: end[b] = te
        heapq.heappush(t_free, (te, s))
    last = sorted(end, key=lambda b: -end[b])[:10]
    for b in last: print(b, 'start', start[b], 'end', end[b], 'tot', tot[b], 'waits', waits[b], 'maxlvl', level(b, len(omega[b]) - 1))
    return max(end.values())
sim2t(o_max, lambda b, d: level(b, d))
nl = sum(1 for b in range(B) if len(omega[b]) > 1 and omega[b][1] > 1.5)
print('problems with omega raised at trip 1:', nl)
print('--- P5: level>=1 lists preempt fresh and run to the end; level 0 probes `probe` slices, then to the end after fresh')
def sim5(order, probe, S=1024, hi_first=True):
    import itertools
    cnt = itertools.count()
    fresh = list(order); fi = 0
    lists = [[] for _ in range(16)]
    pending = []
    t_free = [(0.0, s) for s in range(S)]
    heapq.heapify(t_free)
    done_t = 0.0
    while t_free:
        t, s = heapq.heappop(t_free)
        while pending and pending[0][0] <= t:
            _, _, b, d, v = heapq.heappop(pending); lists[level(b, d)].append((b, d, v))
        job = None
        if hi_first:
            for L in range(15, 0, -1):
                if lists[L]: b, d, v = lists[L].pop(0); job = (b, d, v, True); break
        if job is None and fi < len(fresh):
            job = (fresh[fi], 0, 0, False); fi += 1
        if job is None:
            for L in range(15, -1, -1):
                if lists[L]: b, d, v = lists[L].pop(0); job = (b, d, v, L >= 1 or v >= probe); break
        if job is None:
            if pending: heapq.heappush(t_free, (pending[0][0], s))
            continue
        b, d, v, to_end = job
        nt = len(cost[b])
        run = nt - d if to_end else 1
        te = t + cost[b][d:d + run].sum()
        if d + run < nt: heapq.heappush(pending, (te, next(cnt), b, d + run, v + 1))
        done_t = max(done_t, te)
        heapq.heappush(t_free, (te, s))
    return done_t
for pr in (1, 2, 3):
    print('probe', pr, 'id order', sim5(ids, pr), 'fmax order', sim5(o_max, pr), ' (no preempt: ', sim5(ids, pr, hi_first=False), ')')
print('--- P6: as P5 (fmax order, hi lists preempt + run to end) but level-0 problems continue in slices of q trips (FIFO requeue)')
def sim6(order, probe, q, S=1024, hiq=None):
    import itertools
    cnt = itertools.count()
    fresh = list(order); fi = 0
    lists = [[] for _ in range(16)]
    pending = []
    t_free = [(0.0, s) for s in range(S)]
    heapq.heapify(t_free)
    done_t = 0.0
    while t_free:
        t, s = heapq.heappop(t_free)
        while pending and pending[0][0] <= t:
            _, _, b, d, v = heapq.heappop(pending); lists[level(b, d)].append((b, d, v))
        job = None
        for L in range(15, 0, -1):
            if lists[L]: b, d, v = lists[L].pop(0); job = (b, d, v, hiq or 1000); break
        if job is None and fi < len(fresh):
            job = (fresh[fi], 0, 0, 1); fi += 1
        if job is None:
            if lists[0]: b, d, v = lists[0].pop(0); job = (b, d, v, 1 if v < probe else q)
        if job is None:
            if pending: heapq.heappush(t_free, (pending[0][0], s))
            continue
        b, d, v, run = job
        nt = len(cost[b])
        run = min(run, nt - d)
        te = t + cost[b][d:d + run].sum()
        if d + run < nt: heapq.heappush(pending, (te, next(cnt), b, d + run, v + 1))
        done_t = max(done_t, te)
        heapq.heappush(t_free, (te, s))
    return done_t
for q in (1, 2, 3, 4, 6, 8, 1000):
    print('q', q, 'probe1', sim6(o_max, 1, q), 'probe2', sim6(o_max, 2, q), 'id order probe 1:', sim6(ids, 1, q), 'hi also sliced q:', sim6(o_max, 1, q, hiq=q))
print('--- fmax order with the CURRENT policy:', sim(o_max, 2), 'probe1', sim(o_max, 1), 'probe3', sim(o_max, 3))
# feature variants for ordering under P5
for name, f in [('max', f_max), ('sum', f_sum), ('max+0.02sum', f_max + 0.02 * f_sum)]:
    o = np.argsort(-f, kind='stable')
    print(name, 'current policy', sim(o, 2), 'P5', sim5(o, 2))
