"""The C ABI's process model (-m gpu): one host process, several threads, several devices (include/kvz_hip.h "Devices and threads") -- what kvazaar's tile threads
need to put tile i on GPU i % count (encoderstate.c:944-1013 builds a sub-encoder per tile, threadqueue.c:275-355 runs their jobs on the workers of one process);
struct versioning of the two parameter structs; a worker thread that exits gives its streams and scratch back."""
import ctypes as C
import threading

import numpy as np
import pytest

import ctu_common as cc
import flatapi
import inter_common as ic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import kvazaar_amd
    kvazaar_amd.load_library()
    lib = C.CDLL(kvazaar_amd.LIB_PATH, mode=C.RTLD_GLOBAL)
    assert lib.kvz_hip_device_count() >= 1
    return lib


@pytest.fixture(scope="module")
def oracle():
    return flatapi.load_oracle()


def test_eight_tile_threads_each_on_their_own_device(hip, oracle):
    """Eight host threads of ONE process, thread i with a batch of its own on device i % kvz_hip_device_count() (all on device 0 of a one-GPU box, eight devices of an
    eight-GPU node), all running at once: every thread's pictures must be the oracle's.  Each thread also makes a per-call strategy call and a device-pointer call,
    which must land on the device its batch bound the thread to."""
    from kvazaar_amd.batch import HipBatch, cost_model
    n_dev = hip.kvz_hip_device_count()
    hip.kvz_hip_thread_device.restype = C.c_int
    model = cost_model(hip, 22)
    w, h = 128, 64
    frames = [cc.yuv_frames(w, h, 2, 100 + i, "small") for i in range(8)]
    want = [[cc.run_oracle(oracle, model, w, h, f) for f in frames[i]] for i in range(8)]
    errors, barrier = [], threading.Barrier(8)
    flat = flatapi.FlatLib(hip._name, "kvz_hip_")

    def tile_thread(i):
        try:
            dev = i % n_dev
            batch = HipBatch(hip, w, h, 2, device=dev)
            assert hip.kvz_hip_thread_device() == dev  # the batch bound this thread
            for k, f in enumerate(frames[i]):
                batch.upload(k, f)
            barrier.wait()
            batch.run(model)
            for k in range(2):
                diff = cc.compare(batch.download(k), want[i][k])
                assert not diff, (i, k, diff)
            a = np.arange(256, dtype=np.uint8)
            b = np.roll(a, 3 + i)
            assert flat.satd_nxn(16, flatapi.ptr(a), flatapi.ptr(b)) == oracle.satd_nxn(16, flatapi.ptr(a), flatapi.ptr(b))  # per-call path on this thread's device
            assert hip.kvz_hip_thread_device() == dev
            batch.close()
        except BaseException as e:  # noqa: BLE001
            errors.append((i, repr(e)))
            try:
                barrier.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=tile_thread, args=(i,)) for i in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_set_thread_device_and_unknown_device(hip):
    hip.kvz_hip_thread_device.restype = C.c_int
    n = hip.kvz_hip_device_count()
    assert hip.kvz_hip_set_thread_device(n) == 0 and hip.kvz_hip_set_thread_device(-1) == 0  # refused, the thread keeps its device
    got = []

    def worker():
        assert hip.kvz_hip_set_thread_device(n - 1) == 1
        got.append(hip.kvz_hip_thread_device())
    t = threading.Thread(target=worker)
    t.start()
    t.join()
    assert got == [n - 1]
    hip.kvz_hip_batch_create_on.restype = C.c_void_p
    assert not hip.kvz_hip_batch_create_on(n, 64, 64, 1)  # no such device: NULL, no abort


def test_structs_of_another_size_are_refused(hip):
    """struct_size (include/kvz_hip_types.h, kvz_hip_dev.h): a caller compiled against other headers is told so (-1) instead of being read past its struct"""
    from kvazaar_amd.batch import BatchError, CostModel, HipBatch, cost_model
    from kvazaar_amd import inter
    model = cost_model(hip, 22)
    assert model.struct_size == C.sizeof(CostModel)
    batch = HipBatch(hip, 64, 64, 1)
    batch.upload(0, cc.yuv_frames(64, 64, 1, 5, "small")[0])
    for bad in (0, C.sizeof(CostModel) - 4, C.sizeof(CostModel) + 8):
        m = CostModel.from_buffer_copy(bytes(model))
        m.struct_size = bad
        assert batch.launch(m) == -1
        with pytest.raises(BatchError):
            batch.entropy_code(m)
    batch.run(model)  # the batch is still usable
    batch.close()
    ip = inter.InterPictures(hip, 64, 64, 1)
    clip = ic.clip(64, 64, 2, 3)
    ip.upload(0, clip[1], clip[0], inter.intra_picture_cu_info(64, 64))
    for bad in (0, C.sizeof(inter.InterParams) - 4, C.sizeof(inter.InterParams) + 4):
        p = inter.veryfast_params(22, 1)
        p.struct_size = bad
        with pytest.raises(RuntimeError):
            ip.run(p)
    ip.run(inter.veryfast_params(22, 1))
    ip.close()


def test_worker_threads_give_their_device_memory_back(hip):
    """Thread churn: every worker thread makes per-call strategy calls (its stream + 1 MiB pinned and 1 MiB device arena) and an inter CTU pass (slabs, contexts, ticket
    lists in its thread-local scratch) and exits.  Free device memory must be flat over the generations: a thread that exits returns what it allocated lazily."""
    import kvazaar_amd  # noqa: F401
    from kvazaar_amd import inter
    free_of = C.c_size_t()
    total = C.c_size_t()
    rt = C.CDLL("libamdhip64.so")

    def free_bytes():
        assert rt.hipMemGetInfo(C.byref(free_of), C.byref(total)) == 0
        return free_of.value
    flat = flatapi.FlatLib(hip._name, "kvz_hip_")
    clip = ic.clip(128, 128, 2, 9)

    def worker():
        a = np.arange(256, dtype=np.uint8)
        flat.satd_nxn(16, flatapi.ptr(a), flatapi.ptr(a[::-1].copy()))
        ip = inter.InterPictures(hip, 128, 128, 4)
        for s in range(4):
            ip.upload(s, clip[1], clip[0], inter.intra_picture_cu_info(128, 128))
        ip.run(inter.veryfast_params(22, 1))
        ip.close()

    def generation(n):
        for _ in range(n):
            t = threading.Thread(target=worker)
            t.start()
            t.join()
    generation(3)  # warm: the process-wide allocations (tables, entropy scratch, module load)
    hip.kvz_hip_dev_sync()
    before = free_bytes()
    generation(24)
    after = free_bytes()
    assert before - after < 4 << 20, f"{(before - after) / 2**20:.1f} MiB of device memory lost over 24 worker threads"
