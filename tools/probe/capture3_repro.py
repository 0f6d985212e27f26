"""Minimal reproducer attempt for the `hipStreamEndCapture` crash with THREE capturing streams: no
deft_amd code, only torch -- a launch list of ~90 small kernels spread over S streams with cross-stream event edges (fork from and join
to the capture stream, events between branches), captured into a hipGraph, replayed, destroyed; repeated.  Each (S, repetition) runs
in its own process so that a segfault is counted, not fatal:

    python tools/probe/capture3_repro.py [reps=30]      ->  per S: captures survived / crashed (signal)
"""
import subprocess
import sys


CHILD = r'''
import sys, torch
S, seed = int(sys.argv[1]), int(sys.argv[2])
torch.manual_seed(seed)
dev = torch.device("cuda")
bufs = [torch.zeros(1 << 16, device=dev) for _ in range(12)]
import random
rnd = random.Random(seed)
N = 90
where = [0] + [rnd.randrange(S) for _ in range(N - 1)]
deps = [[] if i == 0 else sorted(set(rnd.sample(range(max(0, i - 6), i), k=min(i, rnd.randint(1, 2))))) for i in range(N)]
def run(streams, events):
    main = torch.cuda.current_stream()
    fork = torch.cuda.Event(); fork.record(main)
    for s in streams[1:]:
        s.wait_event(fork)
    for i in range(N):
        st = main if where[i] == 0 else streams[where[i]]
        for j in deps[i]:
            if where[j] != where[i]:
                st.wait_event(events[j])
        with torch.cuda.stream(st):
            bufs[i % 12].add_(1.0)
        events[i].record(st)
    for s in streams[1:]:
        e = torch.cuda.Event(); e.record(s); main.wait_event(e)
for rep in range(3):
    streams = [None] + [torch.cuda.Stream() for _ in range(S - 1)]
    events = [torch.cuda.Event() for _ in range(N)]
    g = torch.cuda.CUDAGraph()
    cap = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=cap):
        run(streams, events)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    del g
print("ok")
'''


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    for S in (2, 3, 4):
        ok, bad = 0, {}
        for r in range(reps):
            p = subprocess.run([sys.executable, "-c", CHILD, str(S), str(r)], capture_output=True, text=True, timeout=120)
            if p.returncode == 0 and "ok" in p.stdout:
                ok += 1
            else:
                bad[p.returncode] = bad.get(p.returncode, 0) + 1
                last = (p.stderr or "")[-300:]
        print("streams %d: %d of %d processes (3 captures each) survived; exit codes of the others: %s%s" % (S, ok, reps, bad, ("  last stderr: " + last.replace("\n", " | ")) if bad else ""), flush=True)


if __name__ == "__main__":
    main()
