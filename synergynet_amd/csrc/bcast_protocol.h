// The control flow of syn_bcast_constants (include/synergy_hip.h), apart from HIP and RCCL so that it can be compiled for the host and
// run with two threads and a fake transport (tests/bcast_protocol_harness.cpp, tests/test_bcast_protocol_cpu.py).
// Reference construct it replaces: every rank of benchmark.py:112 / main_train.py:176 loads the checkpoint and the basis files itself.
//
// Rule (VERDICT r4 / ADVICE r4): A COLLECTIVE FAILS COLLECTIVELY.  A rank never returns between two collectives the other ranks are going
// to enter; whatever can fail on one rank alone (nothing loaded, an allocation, the export, the import) is done BEFORE a collective that
// carries its verdict to everybody:
//
//   root:      size = constants_bytes(); stage = alloc(size); export(stage)        -- any failure: word = {0, code}
//   all:       broadcast(word)                                                      -- collective 1
//   all:       word.size == 0 -> every rank returns word.code (the root's error), no further collective
//   non-root:  stage = alloc(word.size)                                             -- failure: local code
//   all:       code = agree(local code)                                             -- collective 2 (max over ranks); != 0 -> every rank returns it
//   all:       broadcast(stage, word.size)                                          -- collective 3
//   non-root:  import(stage)
//   all:       return agree(local code)                                             -- collective 4: every rank returns the same verdict
//
// `Ops` supplies the pieces: int rank(); bool loaded(); uint64 bytes(); void *alloc(uint64) (nullptr on failure); void release(void *);
// int export_to(void *, uint64); int import_from(void *, uint64); int broadcast_word(BcastWord *host_word, int root) (the 16-byte size word, which
// lives in HOST memory) and int broadcast(void *blob, uint64 bytes, int root) (the blob, in whatever memory alloc() returns) -- two calls, so
// that no transport has to guess from a byte count which kind of pointer it was handed; transport error -> != 0;
// int agree(int code) (the most severe = most negative code over all ranks; a transport without a reduction returns `code` unchanged).
#pragma once
#include <cstdint>

namespace syn {

struct BcastWord {
    uint64_t size;          // bytes of the blob; 0 = the root could not produce one
    int64_t code;           // the root's error code when size == 0
};

constexpr uint64_t kBcastMinBytes = 256 /* = sizeof(ConstHeader): no blob is smaller than its header */, kBcastMaxBytes = 1ull << 34;

// returns 0 or the (negative) error code EVERY rank of the communicator returns; *stage_out: the rank's copy of the blob (caller releases)
template <class Ops>
int bcast_constants_protocol(Ops &ops, int root, int err_not_loaded, int err_alloc, int err_transport, int err_invalid, const char **where) {
    const bool is_root = ops.rank() == root;
    BcastWord w{0, 0};
    void *stage = nullptr;
    *where = "";
    if (is_root) {
        if (!ops.loaded()) { w.code = err_not_loaded; *where = "the root handle has loaded nothing"; }
        else {
            w.size = ops.bytes();
            stage = ops.alloc(w.size);
            if (!stage) { w.size = 0; w.code = err_alloc; *where = "staging allocation on the root"; }
            else if (int rc = ops.export_to(stage, w.size)) { w.size = 0; w.code = rc; *where = "export on the root"; }
        }
    }
    if (ops.broadcast_word(&w, root)) {                      // collective 1 (a transport error is the communicator's: nothing to agree on)
        if (stage) ops.release(stage);
        *where = "broadcast of the size word";
        return err_transport;
    }
    if (w.size == 0) {                                             // every rank sees the root's verdict and leaves together
        if (stage) ops.release(stage);
        if (!is_root) *where = "the root reported a failure";
        return w.code ? (int)w.code : err_not_loaded;
    }
    int local = 0;
    if (w.size < kBcastMinBytes || w.size > kBcastMaxBytes) { local = err_invalid; *where = "implausible blob size"; }
    else if (!is_root) {
        stage = ops.alloc(w.size);
        if (!stage) { local = err_alloc; *where = "staging allocation"; }
    }
    const int common = ops.agree(local);                           // collective 2
    if (common) {
        if (stage) ops.release(stage);
        if (!local) *where = "another rank could not stage the blob";
        return common;
    }
    if (ops.broadcast(stage, w.size, root)) {                      // collective 3
        ops.release(stage);
        *where = "broadcast of the blob";
        return err_transport;
    }
    if (!is_root) {
        local = ops.import_from(stage, w.size);
        if (local) *where = "import";
    }
    ops.release(stage);
    const int verdict = ops.agree(local);                          // collective 4
    if (verdict && !local) *where = "another rank could not import the blob";
    return verdict;
}

}  // namespace syn
