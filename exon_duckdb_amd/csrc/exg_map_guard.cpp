// exg_map_guard.cpp — see exg_map_guard.hpp.  Host only.
#include "exg_map_guard.hpp"

#include <signal.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>

namespace exg_rd {
namespace {

constexpr int kSlots = 4096;  // mappings alive at once (a reader holds one per open text file + the chunks still out); full: add() = -1 and the reader refuses the file
struct Slot {
    std::atomic<uintptr_t> base{0};  // 0 = free; 1 = being filled
    std::atomic<size_t> len{0};
    std::atomic<uint32_t> hits{0};
};
Slot g_slots[kSlots];
std::atomic<uint64_t> g_patched{0};
std::atomic<int> g_installed{0};  // 0 no, 1 in progress, 2 yes
struct sigaction g_prev;
long g_page = 4096;

void on_sigbus(int sig, siginfo_t *si, void *uctx) {
    const uintptr_t a = (uintptr_t)si->si_addr;
    for (int i = 0; i < kSlots; i++) {
        const uintptr_t b = g_slots[i].base.load(std::memory_order_acquire);
        if (b <= 1) continue;
        const size_t n = g_slots[i].len.load(std::memory_order_acquire);
        if (a < b || a - b >= n) continue;
        // a page of the file that is no longer there: zeros from here on (the reader reports EXG_E_IO at its next call)
        const uintptr_t page = a & ~(uintptr_t)(g_page - 1);
        if (mmap((void *)page, (size_t)g_page, PROT_READ, MAP_FIXED | MAP_PRIVATE | MAP_ANONYMOUS, -1, 0) == MAP_FAILED) break;
        g_slots[i].hits.fetch_add(1, std::memory_order_release);
        g_patched.fetch_add(1, std::memory_order_relaxed);
        return;  // the faulting access restarts
    }
    // not ours: whoever was there before
    // (sa_handler and sa_sigaction share their storage: a disposition of SIG_IGN / SIG_DFL is one whatever SA_SIGINFO says)
    if (g_prev.sa_handler == SIG_IGN) {
        return;
    } else if (g_prev.sa_handler == SIG_DFL || g_prev.sa_handler == nullptr) {
        // (falls through to the default action below)
    } else if (g_prev.sa_flags & SA_SIGINFO) {
        g_prev.sa_sigaction(sig, si, uctx);
        return;
    } else if (g_prev.sa_handler == SIG_IGN) {
        return;
    } else if (g_prev.sa_handler != SIG_DFL && g_prev.sa_handler != nullptr) {
        g_prev.sa_handler(sig);
        return;
    }
    // default action: restore it and return — the access faults again and the process ends the way it would have
    struct sigaction dfl;
    memset(&dfl, 0, sizeof dfl);
    dfl.sa_handler = SIG_DFL;
    sigemptyset(&dfl.sa_mask);
    sigaction(SIGBUS, &dfl, nullptr);
}

void install_once() {
    int want = 0;
    if (g_installed.load(std::memory_order_acquire) == 2) return;
    if (g_installed.compare_exchange_strong(want, 1)) {
        g_page = sysconf(_SC_PAGESIZE) > 0 ? sysconf(_SC_PAGESIZE) : 4096;
        struct sigaction sa;
        memset(&sa, 0, sizeof sa);
        sa.sa_sigaction = on_sigbus;
        sa.sa_flags = SA_SIGINFO | SA_NODEFER | SA_ONSTACK;
        sigemptyset(&sa.sa_mask);
        memset(&g_prev, 0, sizeof g_prev);
        sigaction(SIGBUS, &sa, &g_prev);
        g_installed.store(2, std::memory_order_release);
    } else {
        while (g_installed.load(std::memory_order_acquire) != 2) {
        }
    }
}

}  // namespace

int MapGuard::add(const void *base, size_t len) {
    if (!base || !len) return -1;
    install_once();
    for (int i = 0; i < kSlots; i++) {
        uintptr_t free_ = 0;
        if (g_slots[i].base.compare_exchange_strong(free_, 1)) {
            g_slots[i].hits.store(0, std::memory_order_relaxed);
            g_slots[i].len.store(len, std::memory_order_release);
            g_slots[i].base.store((uintptr_t)base, std::memory_order_release);
            return i;
        }
    }
    return -1;
}

void MapGuard::remove(int slot) {
    if (slot < 0 || slot >= kSlots) return;
    g_slots[slot].len.store(0, std::memory_order_release);
    g_slots[slot].base.store(0, std::memory_order_release);
}

bool MapGuard::hit(int slot) {
    if (slot < 0 || slot >= kSlots) return false;
    return g_slots[slot].hits.load(std::memory_order_acquire) != 0;
}

uint64_t MapGuard::patched() { return g_patched.load(std::memory_order_relaxed); }

}  // namespace exg_rd
