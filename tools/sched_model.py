"""Scheduling models of a launch of one-wave workgroups (analysis only): list scheduling onto independent slots, and processor sharing
inside a SIMD with the measured issue rates.  Used by tools/gpu_tail_model.py and tools/cpu_forward_order_model.py."""
import heapq


def makespan(jobs, S):
    """greedy list scheduling in the given order on S identical slots (what the hardware dispatcher does with a launch)"""
    if len(jobs) <= S:
        return float(max(jobs)) if len(jobs) else 0.0
    h = [0.0] * S
    heapq.heapify(h)
    end = 0.0
    for j in jobs:
        t = heapq.heappop(h) + float(j)
        end = max(end, t)
        heapq.heappush(h, t)
    return end


# aggregate issue rate of a SIMD with n resident waves of the blend body, relative to 8 waves (profiles/r03_valu_issue.txt,
# "forward blend body ... (product form)": 12.2 / 28.2 / - / 39.7 / - / 43.6 / - / 46.6 T lane-ops/s; odd counts interpolated)
F = [0.0, 0.263, 0.605, 0.74, 0.852, 0.90, 0.936, 0.97, 1.0]

def ps_makespan(jobs, nsimd, K):
    """waves dispatched in order to the SIMD with the fewest resident waves (free slot), processor sharing inside a SIMD"""
    jobs = [float(j) for j in jobs]
    n = len(jobs)
    rem = [[] for _ in range(nsimd)]      # per SIMD: remaining work of resident waves
    tlast = [0.0] * nsimd
    nxt = 0
    now = 0.0
    # initial fill, round-robin
    for k in range(K):
        for s in range(nsimd):
            if nxt < n:
                rem[s].append(jobs[nxt]); nxt += 1
    def next_finish(s):
        r = rem[s]
        if not r: return None
        rate = F[len(r)] / len(r)
        return tlast[s] + min(r) / rate
    heap = []
    for s in range(nsimd):
        t = next_finish(s)
        if t is not None: heapq.heappush(heap, (t, s, len(rem[s]), 0))
    ver = [0] * nsimd
    end = 0.0
    while heap:
        t, s, cnt, v = heapq.heappop(heap)
        if v != ver[s]: continue
        r = rem[s]
        rate = F[len(r)] / len(r)
        dt = t - tlast[s]
        w = dt * rate
        r2 = [x - w for x in r]
        mn = min(r2)
        r2.remove(mn)
        rem[s] = r2
        tlast[s] = t
        end = max(end, t)
        if nxt < n:
            rem[s].append(jobs[nxt]); nxt += 1
        ver[s] += 1
        tn = next_finish(s)
        if tn is not None: heapq.heappush(heap, (tn, s, len(rem[s]), ver[s]))
    return end
