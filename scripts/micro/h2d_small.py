import time, torch
dev = torch.device("cuda:0")
x = torch.randn(4096, 4096, device=dev)
def t(f, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
print("torch.tensor(list, device)       ", t(lambda: torch.tensor([1, 2, 3, 4], dtype=torch.int32, device=dev)))
print("torch.tensor(list).to(dev)       ", t(lambda: torch.tensor([1, 2, 3, 4], dtype=torch.int32).to(dev)))
print("pinned + non_blocking            ", t(lambda: torch.tensor([1, 2, 3, 4], dtype=torch.int32).pin_memory().to(dev, non_blocking=True)))
buf = torch.empty(4, dtype=torch.int32, device=dev)
print("copy_ into device buf (pageable) ", t(lambda: buf.copy_(torch.tensor([1, 2, 3, 4], dtype=torch.int32))))
print("copy_ non_blocking (pageable)    ", t(lambda: buf.copy_(torch.tensor([1, 2, 3, 4], dtype=torch.int32), non_blocking=True)))
print("torch.full                       ", t(lambda: torch.full((2,), 6561, dtype=torch.int64, device=dev)))
print("torch.arange device              ", t(lambda: torch.arange(100, dtype=torch.int32, device=dev)))
print("x.cpu() small                    ", t(lambda: buf.cpu()))
