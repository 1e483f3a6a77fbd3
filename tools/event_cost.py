"""What does an event record / a cross-stream wait cost the stream it sits in?  (eager fork-join anatomy)"""
import time, torch
x = torch.zeros(1 << 16, device="cuda")
s1, s2 = torch.cuda.current_stream(), torch.cuda.Stream()
def run(mode, n=400):
    evs = [torch.cuda.Event() for _ in range(n)]
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        x.add_(1.0)
        if mode >= 1:
            evs[i].record(s1)
        if mode >= 2:
            s2.wait_event(evs[i])
        if mode >= 3:
            with torch.cuda.stream(s2):
                x2.add_(1.0)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
x2 = torch.zeros(1 << 16, device="cuda")
for m, name in enumerate(["kernels only", "+ event record on the stream", "+ other stream waits on it", "+ kernel on the other stream"]):
    run(m, 50)
    print(f"{name:36s} {run(m):6.2f} us per iteration on the main stream")
# join cost: main waits on an event of the other stream every iteration
def run_join(n=400):
    evs = [torch.cuda.Event() for _ in range(n)]
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        x.add_(1.0)
        with torch.cuda.stream(s2):
            x2.add_(1.0)
            evs[i].record(s2)
        s1.wait_event(evs[i])
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
run_join(50)
print(f"{'main waits for the other stream':36s} {run_join():6.2f} us per iteration")
