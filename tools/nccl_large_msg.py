#!/usr/bin/env python3
"""Where do large all_to_all_single messages break over nccl (= RCCL)?  One rank, one GPU: the message is a self-copy.

For a list of message sizes around 2 GiB and 4 GiB, in 8-byte and 1-byte elements, with and without explicit split sizes:
fill the source with a position-dependent pattern, the destination with a sentinel, run the collective once and report
the first byte offset that differs, how many bytes differ, and whether the destination still holds the sentinel there
(= never written) or something else (= written with the wrong data).  The result decides multi_gpu.MAX_MESSAGE_BYTES.
"""
import os
import sys

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29547")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
import torch
import torch.distributed as dist

torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
GiB = 1 << 30
sizes = [GiB, 2 * GiB - 4096, 2 * GiB, 2 * GiB + 4096, 3 * GiB, 4 * GiB - 4096, 4 * GiB, 4 * GiB + 4096, 7 * GiB]
if len(sys.argv) > 1:
    sizes = [int(float(x) * GiB) for x in sys.argv[1:]]


def pattern(n_bytes, dtype):
    if dtype == torch.int64:
        return torch.arange(n_bytes // 8, dtype=torch.int64, device="cuda") * 0x9E3779B97F4A7C15 + 12345
    i = torch.arange(n_bytes, dtype=torch.int64, device="cuda")
    return ((i * 2654435761) >> 7).to(torch.uint8)


def report(name, src, out, sentinel):
    es, step = src.element_size(), 1 << 27
    n_bad = untouched = 0
    first = last = -1
    for lo in range(0, src.numel(), step):  # in pieces: nonzero() of 2^31 elements does not fit
        a, b = src[lo:lo + step], out[lo:lo + step]
        diff = a != b
        nb = int(diff.sum().item())
        if not nb:
            continue
        idx = torch.nonzero(diff, as_tuple=False).view(-1)
        if first < 0:
            first = lo + int(idx[0].item())
        last = lo + int(idx[-1].item())
        n_bad += nb
        untouched += int((b[idx] == sentinel).sum().item())
    if n_bad == 0:
        print(f"  {name:28s} exact", flush=True)
        return
    print(f"  {name:28s} DAMAGED: {n_bad * es} bytes differ, first at byte {first * es} (= {first * es / GiB:.6f} GiB), "
          f"last at byte {last * es + es - 1}; {untouched * es} of them still hold the sentinel (never written), "
          f"{(n_bad - untouched) * es} hold other data", flush=True)


for n_bytes in sizes:
    print(f"message of {n_bytes} bytes ({n_bytes / GiB:.6f} GiB)", flush=True)
    for dtype, sentinel in ((torch.int64, -7), (torch.uint8, 0xA5)):
        src = pattern(n_bytes, dtype)
        n = src.numel()
        for splits in (True, False):
            out = torch.full_like(src, sentinel)
            if splits:
                dist.all_to_all_single(out, src, [n], [n])
            else:
                dist.all_to_all_single(out, src)
            torch.cuda.synchronize()
            report(f"{str(dtype).split('.')[-1]}, {'split sizes' if splits else 'equal split'}", src, out, sentinel)
            del out
        del src
        torch.cuda.empty_cache()
dist.destroy_process_group()
