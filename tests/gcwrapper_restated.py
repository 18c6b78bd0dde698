"""TEST INFRASTRUCTURE: a compact restatement of the Python half of the reference's batch interface,
    src_py/elf/utils_elf.py   Allocator._alloc :31-57, Allocator.spec2batches :59-99, GCWrapper :291-437 (reg_callback :340-359,
                              _call :368-414, run :426-437), Batch :112-290 (only what _call uses)
for the GPU box, where /root/reference does not exist.  tests/test_pybind_boundary.py drives the pybind11 boundary (_elf,
_elfgames_go*) with the reference's own unmodified file when it is present, and checks on the CPU that this restatement makes the
same calls in the same order with the same arguments (recorded on a mock context), so the GPU run through it stands for a run
through the original.  One extension: device_resident=True allocates the tensors in HBM instead of (pinned) host memory."""
from collections import defaultdict

import torch

TORCH_TYPES = {"int32_t": torch.int32, "int64_t": torch.int64, "float": torch.float32, "unsigned char": torch.uint8, "char": torch.uint8}


class Batch:
    def __init__(self, GC=None, tensors=None):
        self.GC, self.batch = GC, dict(tensors or {})

    def __getitem__(self, key):
        if key in self.batch:
            return self.batch[key]
        if "last_" + key in self.batch:
            return self.batch["last_" + key][1:]
        raise KeyError("Batch(): specified key: %s or %s not found!" % (key, "last_" + key))

    def __contains__(self, key):
        return key in self.batch or "last_" + key in self.batch

    def first_k(self, k):
        return Batch(self.GC, {name: t[:k] for name, t in self.batch.items()})

    def to(self, gpu):
        return Batch(self.GC, {name: t.cuda(gpu, non_blocking=True) for name, t in self.batch.items()})


def alloc_field(p, gpu, device_resident):
    f = p.field()
    t = torch.empty(tuple(f.sz().vec()), dtype=TORCH_TYPES[f.type_name()], device=("cuda:%d" % gpu) if device_resident else "cpu")
    if gpu is not None and not device_resident:
        with torch.cuda.device(gpu):
            t = t.pin_memory()
    t.fill_(1)
    p.set(t.data_ptr(), [st * t.element_size() for st in t.stride()])
    return f.name(), t


def spec2batches(ctx, batchsize, spec, gpu, num_recv=1, device_resident=False):
    batches, name2idx, idx2name = [], defaultdict(list), {}
    for name, v in spec.items():
        v["input"] = v.get("input") or []
        v["reply"] = v.get("reply") or []
        keys = list(set(v["input"] + v["reply"]))
        opts = ctx.createSharedMemOptions(name, v.get("batchsize", batchsize))
        opts.setTimeout(v.get("timeout_usec", 0))
        for _ in range(num_recv):
            smem = ctx.allocateSharedMem(opts, keys)
            tensors = dict(alloc_field(smem[k], gpu, device_resident) for k in keys)
            batches.append(dict(input={k: tensors[k] for k in v["input"]}, reply={k: tensors[k] for k in v["reply"]}))
            idx = smem.getSharedMemOptions().idx()
            name2idx[name].append(idx)
            idx2name[idx] = name
    return batches, name2idx, idx2name


class GCWrapper:
    def __init__(self, GC, batchsize, spec, gpu=None, params=None, num_recv=1, device_resident=False, **_):
        self.GC, self.gpu, self.params, self.device_resident = GC, gpu, params or {}, device_resident
        self.batches, self.name2idx, self.idx2name = spec2batches(GC.ctx(), batchsize, spec, gpu, num_recv, device_resident)
        self._cb = {}

    def reg_has_callback(self, key):
        return key in self.name2idx

    def reg_callback_if_exists(self, key, cb):
        if self.reg_has_callback(key):
            self.reg_callback(key, cb)
            return True
        return False

    def reg_callback(self, key, cb):
        if key not in self.name2idx:
            raise ValueError("Callback[%s] is not in the specification" % key)
        for idx in self.name2idx[key]:
            self._cb[idx] = cb
        return True

    def _call(self, smem):
        idx = smem.getSharedMemOptions().idx()
        if idx not in self._cb:
            raise ValueError("smem.idx[%d] is not in callback functions" % idx)
        if self._cb[idx] is None:
            return
        k = smem.effective_batchsize()
        assert k > 0
        picked = Batch(self.GC, self.batches[idx]["input"]).first_k(k)
        if self.gpu is not None and not self.device_resident:
            picked = picked.to(self.gpu)
        picked.smem, picked.batchsize, picked.max_batchsize = smem, k, smem.getSharedMemOptions().batchsize()
        sel_reply = Batch(self.GC, self.batches[idx]["reply"]).first_k(k)
        reply = self._cb[idx](picked)
        if isinstance(reply, dict):
            extra = [key for key in reply if key not in sel_reply.batch]
            missing = [key for key in sel_reply.batch if key not in reply]
            for key, dst in sel_reply.batch.items():
                if key in reply and reply[key] is not None:
                    src = reply[key]
                    if isinstance(src, (int, float)):
                        dst.fill_(src)
                    else:
                        dst[:] = src.squeeze()
            if extra:
                raise ValueError("Receive extra keys %s from reply!" % str(extra))
            if missing:
                raise ValueError("Missing keys %s absent in reply!" % str(missing))

    def start(self):
        for key, indices in self.name2idx.items():
            for idx in indices:
                if idx not in self._cb:
                    raise ValueError("GCWrapper.start(): No callback function for key = %s and idx = %d" % (key, idx))
        self.GC.ctx().start()

    def run(self):
        smem = self.GC.ctx().wait()
        self._call(smem)
        self.GC.ctx().step()

    def stop(self):
        self.GC.ctx().stop()
