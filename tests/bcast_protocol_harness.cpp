// Host-side harness of synergynet_amd/csrc/bcast_protocol.h (the control flow of syn_bcast_constants): N threads = N ranks on a fake transport
// whose collectives TIME OUT when a rank is missing -- which is how the round-4 defect (a root returning before the first broadcast) shows.
// usage: harness <world> <scenario>     prints one line per rank: "rank r rc <code> imported <0|1> where <text>"; exit code 3 = a collective hung.
// scenarios: ok | root_empty | root_alloc | root_export | peer_alloc | peer_import | no_allreduce_peer_alloc
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../synergynet_amd/csrc/bcast_protocol.h"

namespace {
constexpr int ERR_NOT_LOADED = -3, ERR_HIP = -2, ERR_INVALID = -1, ERR_EXPORT = -7, ERR_IMPORT = -8;
std::atomic<bool> hung{false};

struct Transport {                        // a barrier with a payload, per collective call (all ranks call the same sequence -- or hang)
    int world;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    long generation = 0;
    std::vector<unsigned char> payload;
    int reduce = 0;
    bool collective(std::unique_lock<std::mutex> &lk) {
        const long gen = generation;
        if (++arrived == world) { arrived = 0; ++generation; cv.notify_all(); return true; }
        if (!cv.wait_for(lk, std::chrono::seconds(3), [&] { return generation != gen; })) { hung = true; return false; }
        return true;
    }
    int broadcast(void *buf, uint64_t n, bool is_root) {
        std::unique_lock<std::mutex> lk(m);
        if (is_root) payload.assign((unsigned char *)buf, (unsigned char *)buf + n);
        if (!collective(lk)) return 1;
        // second phase so that the payload is read before anybody's next collective overwrites it
        if (!is_root) memcpy(buf, payload.data(), n);
        if (!collective(lk)) return 1;
        return 0;
    }
    int allreduce_max(int v, bool *ok) {
        std::unique_lock<std::mutex> lk(m);
        if (arrived == 0) reduce = v; else reduce = reduce > v ? reduce : v;
        *ok = collective(lk);
        const int r = reduce;
        if (*ok) *ok = collective(lk);
        return r;
    }
};

struct FakeOps {
    Transport &t; int my_rank; int root; std::string scenario; bool has_allreduce;
    bool imported = false;
    int rank() const { return my_rank; }
    bool loaded() const { return !(scenario == "root_empty" && my_rank == root); }
    uint64_t bytes() const { return 4096; }
    void *alloc(uint64_t n) {
        if (scenario == "root_alloc" && my_rank == root) return nullptr;
        if ((scenario == "peer_alloc" || scenario == "no_allreduce_peer_alloc") && my_rank != root && my_rank == 1) return nullptr;
        return malloc(n);
    }
    void release(void *p) { free(p); }
    int export_to(void *p, uint64_t n) { if (scenario == "root_export") return ERR_EXPORT; memset(p, 0x5a, n); return 0; }
    int import_from(void *p, uint64_t n) {
        if (scenario == "peer_import" && my_rank == 1) return ERR_IMPORT;
        for (uint64_t i = 0; i < n; ++i) if (((unsigned char *)p)[i] != 0x5a) return ERR_IMPORT;
        imported = true;
        return 0;
    }
    int broadcast_word(syn::BcastWord *w, int r) { return t.broadcast(w, sizeof *w, my_rank == r); }
    int broadcast(void *buf, uint64_t n, int r) { return buf ? t.broadcast(buf, n, my_rank == r) : (t.broadcast(nullptr, 0, false), 1); }
    int agree(int code) {
        if (!has_allreduce) return code;
        bool ok = true;
        const int v = t.allreduce_max(-code, &ok);
        return ok ? -v : (code ? code : ERR_HIP);
    }
};
}  // namespace

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    const int world = atoi(argv[1]);
    const std::string scenario = argv[2];
    const int root = world > 1 && scenario.rfind("peer", 0) != 0 && scenario.rfind("no_allreduce", 0) != 0 ? world - 1 : 0;    // not always rank 0
    Transport t{world};
    std::vector<std::string> lines(world);
    std::vector<std::thread> th;
    for (int r = 0; r < world; ++r)
        th.emplace_back([&, r] {
            FakeOps ops{t, r, root, scenario, scenario.rfind("no_allreduce", 0) != 0};
            const char *where = "";
            const int rc = syn::bcast_constants_protocol(ops, root, ERR_NOT_LOADED, ERR_HIP, ERR_HIP, ERR_INVALID, &where);
            char buf[256];
            snprintf(buf, sizeof buf, "rank %d rc %d imported %d where %s", r, rc, (int)ops.imported, where);
            lines[r] = buf;
        });
    for (auto &x : th) x.join();
    for (auto &l : lines) puts(l.c_str());
    return hung ? 3 : 0;
}
