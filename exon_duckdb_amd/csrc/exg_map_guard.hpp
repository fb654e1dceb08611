// exg_map_guard.hpp — a file that shrinks while its mapping is being read must be an I/O error, not a dead process.
//
// Plain text inputs are mapped read-only and the DataChunk strings point straight into the mapping (zero-copy payload,
// DESIGN 3): the bytes of a string are read by the CONSUMER (DuckDB's operators), long after exg_next_chunk returned.  When
// another process truncates the file meanwhile, touching a page behind the new end raises SIGBUS — by default the end of the
// DuckDB process; the reference's buffered reader returns an error there (a short read).  The guard makes it one here:
//
//   * every file mapping of a reader is registered (base, length) in a fixed table;
//   * a process-wide SIGBUS handler looks the faulting address up; inside a registered mapping it maps an anonymous zero page
//     over the faulting page (mmap MAP_FIXED: a system call, async-signal-safe), marks the mapping "hit" and returns — the
//     faulting load restarts and reads zeros; anywhere else it hands the signal to the handler that was installed before
//     (or restores the default action and returns, so that the fault is raised again and handled as if this guard did not exist);
//   * the reader asks `hit()` at every batch and at every chunk it hands out: EXG_E_IO "... was truncated while it was read".
//
// The rows a consumer was holding when the file shrank may show zeros where the bytes are gone; the next call fails.  No locks,
// no allocation in the handler; slots are claimed with compare-and-swap.  Host only (tests/host_asan_driver.cpp runs it).
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace exg_rd {

struct MapGuard {
    // registers [base, base + len) -> slot (>= 0), or -1 when the table (4096 mappings) is full: the caller refuses the file
    // (EXG_E_NOMEM) rather than read it unguarded
    static int add(const void *base, size_t len);
    // forgets the slot (call BEFORE munmap)
    static void remove(int slot);
    // a fault inside the slot's mapping was patched with a zero page since add()
    static bool hit(int slot);
    // (tests) faults patched so far, process-wide
    static uint64_t patched();
};

}  // namespace exg_rd
