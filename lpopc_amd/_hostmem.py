"""Keep large host allocations of this process mapped.

glibc serves large malloc requests (numpy arrays, torch CPU tensors) with mmap and returns them with munmap.  Page-locked
registrations of such a range — this library's pin_host registrations when a caller breaks the lifetime contract, and the
HIP runtime's own pinning of pageable memory during hipMemcpy of large arrays — then refer to addresses that are no longer
mapped; during this work a long-running process (the whole -m gpu test session, hundreds of engines and thousands of
pageable copies) ended with a bare SIGABRT inside a later pageable copy about once in a dozen sessions, in this library's
staged copies as well as in torch's.  With the two mallopt settings below freed blocks stay in the heap (mapped, reusable),
so a stale registration always points at valid pages.  Called by tests/conftest.py, bench.py and __graft_entry__.smoke();
a product process that follows the pin_host contract does not need it."""
import ctypes

M_TRIM_THRESHOLD, M_MMAP_MAX = -1, -4


def keep_heap_mapped():
    try:
        libc = ctypes.CDLL("libc.so.6")
        ok = libc.mallopt(M_MMAP_MAX, 0) == 1 and libc.mallopt(M_TRIM_THRESHOLD, 2 ** 31 - 1) == 1
        return bool(ok)
    except Exception:
        return False
