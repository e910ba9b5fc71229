// hipemu scheduler: workgroups run one after the other, their threads as fibers on one OS thread (see hip/hip_runtime.h for
// the model).  TEST INFRASTRUCTURE ONLY.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <sys/mman.h>
#include <vector>

// Fiber switch: callee-saved registers + stack pointer (the fibers share the FP control state; no signal masks -- a
// swapcontext() costs a system call per switch, and an emulated MFMA is 64 switches).
extern "C" void hipemu_switch(void** save_sp, void* const* load_sp);
asm(R"(
.text
.globl hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq (%rsi), %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size hipemu_switch, .-hipemu_switch
)");

namespace hipemu {

struct Barrier {
  int count = 0, gen = 0, expected = 0;
};
// A wave operation is a rendezvous of the lanes that issue the SAME instruction (call site).  Lanes of one wave that sit at
// different sites -- a loop some lanes have left, the two sides of a branch -- are what the hardware runs one after the other
// under partial EXEC masks: they are released group by group (see resolve_divergence).
struct Wave {
  int arrived = 0, live = 0, releases = 0;
  alignas(16) unsigned char slot[64][2][64];           // [lane][parity][bytes]: a lane alternates between its two slots
  int rel_parity[64];                                  // slot a lane used in the operation it was last released from
  int rel_id[64];                                      // which release that was (lanes released together share it)
};
struct Fiber {
  void* sp = nullptr;
  Idx tid;
  int lane = 0, wave = 0;
  bool done = true;
  bool blocked = false;               // waiting inside a wave operation (released by another lane / the scheduler)
  const int* wait_gen = nullptr;      // waiting at the workgroup barrier while *wait_gen == seen_gen
  int seen_gen = 0;
  int nops = 0;                       // wave operations issued so far (parity of the slot in use)
  const char* wait_kind = "";
  void* wait_site = nullptr;
};

Fiber* cur = nullptr;
Idx block_idx, block_dim, grid_dim;
alignas(64) unsigned char dyn_lds[160 * 1024 + 64];

static const size_t STACK_BYTES = 512 * 1024;
static std::vector<Fiber> fibers;
static std::vector<Wave> waves;
static unsigned char* stacks = nullptr;
static size_t stacks_for = 0;
static Barrier block_bar;
static void* sched_sp = nullptr;
static const std::function<void()>* body_fn = nullptr;
static bool in_launch = false;
static size_t cur_nthr = 0;

const Idx& thread_idx() { return cur->tid; }
int lane_id() { return cur->lane; }
bool deposited(int lane) {
  const Wave& w = waves[cur->wave];
  return w.rel_id[lane] == w.rel_id[cur->lane];
}
const unsigned char* peer(int lane) {
  const Wave& w = waves[cur->wave];
  return w.slot[lane][w.rel_parity[lane]];
}

static void yield_to_scheduler() { hipemu_switch(&cur->sp, &sched_sp); }

static void arrive(Barrier& b) {
  const int gen = b.gen;
  if (++b.count >= b.expected) {
    b.count = 0;
    ++b.gen;
    return;
  }
  cur->wait_gen = &b.gen;
  cur->seen_gen = gen;
  yield_to_scheduler();
}

void sync_block() {
  cur->wait_kind = "__syncthreads";
  cur->wait_site = __builtin_return_address(0);
  arrive(block_bar);
}

// releases the lanes of wave `wi` that wait at `site` (all waiting lanes when site == nullptr)
static void release(int wi, void* site) {
  Wave& w = waves[wi];
  const int id = ++w.releases;
  for (size_t t = (size_t)wi * 64; t < (size_t)wi * 64 + 64 && t < cur_nthr; ++t) {
    Fiber& f = fibers[t];
    if (f.done || !f.blocked || (site && f.wait_site != site)) continue;
    f.blocked = false;
    w.rel_parity[f.lane] = (f.nops - 1) & 1;
    w.rel_id[f.lane] = id;
    --w.arrived;
  }
}

void wave_exchange(const void* mine, int bytes) {
  void* const site = __builtin_return_address(0);
  Wave& w = waves[cur->wave];
  if (bytes > 64) {
    fprintf(stderr, "hipemu: exchange of %d bytes per lane\n", bytes);
    abort();
  }
  Fiber* f = cur;
  f->wait_kind = "wave operation";
  f->wait_site = site;
  memcpy(w.slot[f->lane][f->nops & 1], mine, bytes);
  ++f->nops;
  f->blocked = true;
  if (++w.arrived >= w.live) {      // every live lane of the wave is here: one instruction, if they all came from one site
    bool same = true;
    for (size_t t = (size_t)f->wave * 64; t < (size_t)f->wave * 64 + 64 && t < cur_nthr && same; ++t)
      same = fibers[t].done || fibers[t].wait_site == site;
    if (same) {
      release(f->wave, nullptr);
      return;
    }
  }
  yield_to_scheduler();
}

// Every live thread of the workgroup is blocked.  A wave with lanes waiting inside wave operations is executing under partial
// EXEC masks -- a loop's last trips, a guarded reduction, the two sides of a branch: the hardware runs such groups one after
// the other, inner / earlier code first (lanes that left a loop wait at its exit for the others).  The group at the lowest
// code address goes first: loop bodies precede their exits, and groups from the sides of a branch do not care.
static bool resolve_divergence(size_t nwaves) {
  bool any = false;
  for (size_t wi = 0; wi < nwaves; ++wi) {
    if (waves[wi].arrived <= 0) continue;
    void* lo = nullptr;
    for (size_t t = wi * 64; t < wi * 64 + 64 && t < cur_nthr; ++t) {
      const Fiber& f = fibers[t];
      if (!f.done && f.blocked && (!lo || (uintptr_t)f.wait_site < (uintptr_t)lo)) lo = f.wait_site;
    }
    release((int)wi, lo);
    any = true;
  }
  return any;
}

static void fiber_exit() {
  Fiber* f = cur;
  f->done = true;      // (its last deposits stay readable: the other lanes may not have consumed them yet)
  Wave& w = waves[f->wave];
  --w.live;
  // a thread that has left no longer counts at the barriers of those still running
  if (--block_bar.expected > 0 && block_bar.count >= block_bar.expected) {
    block_bar.count = 0;
    ++block_bar.gen;
  }
}

static void fiber_main() {
  (*body_fn)();
  fiber_exit();
  hipemu_switch(&cur->sp, &sched_sp);
  abort();      // a finished fiber is never resumed
}

void launch(dim3 grid, dim3 block, size_t dyn_bytes, const std::function<void()>& body) {
  if (in_launch) {
    fprintf(stderr, "hipemu: nested launch\n");
    abort();
  }
  if (dyn_bytes > 160 * 1024) {
    fprintf(stderr, "hipemu: %zu bytes of dynamic LDS requested (160 KB per workgroup)\n", dyn_bytes);
    abort();
  }
  const size_t nthr = (size_t)block.x * block.y * block.z;
  if (nthr == 0 || nthr > 1024) {
    fprintf(stderr, "hipemu: %zu threads per workgroup\n", nthr);
    abort();
  }
  in_launch = true;
  cur_nthr = nthr;
  if (fibers.size() < nthr) fibers.resize(nthr);
  const size_t nwaves = (nthr + 63) / 64;
  if (waves.size() < nwaves) waves.resize(nwaves);
  if (stacks_for < nthr) {
    if (stacks) munmap(stacks, stacks_for * STACK_BYTES);
    stacks = (unsigned char*)mmap(nullptr, nthr * STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (stacks == MAP_FAILED) {
      perror("hipemu: mmap");
      abort();
    }
    stacks_for = nthr;
  }
  body_fn = &body;
  block_dim = Idx{block.x, block.y, block.z};
  grid_dim = Idx{grid.x, grid.y, grid.z};
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        block_idx = Idx{bx, by, bz};
        block_bar = Barrier();
        block_bar.expected = (int)nthr;
        for (size_t w = 0; w < nwaves; ++w) {
          waves[w].arrived = waves[w].live = waves[w].releases = 0;
          memset(waves[w].slot, 0, sizeof(waves[w].slot));
          for (int l = 0; l < 64; ++l) {
            waves[w].rel_parity[l] = 0;
            waves[w].rel_id[l] = -1;
          }
        }
        for (size_t t = 0; t < nthr; ++t) {
          Fiber& f = fibers[t];
          f.tid = Idx{(unsigned)(t % block.x), (unsigned)((t / block.x) % block.y), (unsigned)(t / ((size_t)block.x * block.y))};
          f.lane = (int)(t & 63);
          f.wave = (int)(t >> 6);
          f.done = false;
          f.blocked = false;
          f.wait_gen = nullptr;
          f.nops = 0;
          f.wait_site = nullptr;
          ++waves[f.wave].live;
          // first switch: six zeroed callee-saved registers, then `ret` into fiber_main with the stack as after a call
          void** top = (void**)(stacks + (t + 1) * STACK_BYTES);
          top[-1] = nullptr;
          top[-2] = (void*)&fiber_main;
          for (int r = 3; r <= 8; ++r) top[-r] = nullptr;
          f.sp = (void*)(top - 8);
        }
        size_t remaining = nthr;
        while (remaining) {
          bool progressed = false;
          for (size_t t = 0; t < nthr; ++t) {
            Fiber& f = fibers[t];
            if (f.done || f.blocked) continue;
            if (f.wait_gen) {
              if (*f.wait_gen == f.seen_gen) continue;
              f.wait_gen = nullptr;
            }
            cur = &f;
            hipemu_switch(&sched_sp, &f.sp);
            progressed = true;
            if (f.done) --remaining;
          }
          if (!progressed) progressed = resolve_divergence(nwaves);
          if (!progressed) {
            fprintf(stderr, "hipemu: deadlock in workgroup (%u,%u,%u): %zu threads wait at a workgroup barrier the others never "
                            "reach\n", bx, by, bz, remaining);
            for (size_t t = 0; t < nthr; ++t)
              if (!fibers[t].done && (t == 0 || fibers[t].wait_site != fibers[t - 1].wait_site || fibers[t - 1].done)) {
                Dl_info info;
                const bool ok = dladdr(fibers[t].wait_site, &info) != 0;
                fprintf(stderr, "  from thread %zu: %s at %p (%s + 0x%zx)\n", t, fibers[t].wait_kind, fibers[t].wait_site,
                        ok && info.dli_sname ? info.dli_sname : "?", ok ? (size_t)((char*)fibers[t].wait_site - (char*)info.dli_fbase) : 0);
              }
            abort();
          }
        }
      }
  cur = nullptr;
  body_fn = nullptr;
  in_launch = false;
}

}  // namespace hipemu

// ---- entry points of comm.hip (RCCL) that the emulated library does not carry -------------------------------------------
#include "../../include/twingan_hip.h"
void tg_set_error(const char* fmt, ...);
extern "C" {
int tg_comm_unique_id_bytes(void) { return 128; }
int tg_comm_unique_id(void*) { tg_set_error("hipemu: no RCCL"); return TG_ENOSUP; }
int tg_comm_init(const void*, int, int, void**) { tg_set_error("hipemu: no RCCL"); return TG_ENOSUP; }
int tg_allreduce(void*, void*, int64_t, int, void*) { tg_set_error("hipemu: no RCCL"); return TG_ENOSUP; }
int tg_comm_destroy(void*) { tg_set_error("hipemu: no RCCL"); return TG_ENOSUP; }
}
