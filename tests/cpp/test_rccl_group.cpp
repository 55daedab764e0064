// CPU test of pedoni_amd/csrc/rccl_group.hpp: whatever fails inside a grouped exchange, the group
// is closed, nothing is issued after the first failure, and the first failure is reported.
// Built and run by tests/test_host_cpu.py (g++, no HIP / RCCL needed).
#include "rccl_group.hpp"

#include <cstdio>
#include <string>
#include <vector>

struct Mock {
    int depth = 0;                 // open groups
    std::vector<std::string> log;
    int fail_at = -1;              // index of the operation that fails (-1: none)
    int issued = 0;
    int start_fails = 0, end_fails = 0;
    int start() { if (start_fails) return 7; ++depth; log.push_back("start"); return 0; }
    int end() { --depth; log.push_back("end"); return end_fails ? 9 : 0; }
    int op(const char* name) { log.push_back(name); return issued++ == fail_at ? 5 : 0; }
};

static int failures = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); ++failures; } } while (0)

static GroupOutcome<int> exchange(Mock& m)
{
    return run_group<int>(0, [&] { return m.start(); }, [&] { return m.end(); }, [&](auto&& op) {
        op([&] { return m.op("send_down"); }, "send_down");
        op([&] { return m.op("recv_below"); }, "recv_below");
        op([&] { return m.op("send_up"); }, "send_up");
        op([&] { return m.op("recv_above"); }, "recv_above");
    });
}

int main()
{
    {   // all good
        Mock m; auto o = exchange(m);
        CHECK(o.ok && m.depth == 0 && m.issued == 4 && m.log.size() == 6 && m.log.back() == "end");
    }
    for (int k = 0; k < 4; ++k) {   // operation k fails: closed, nothing after it, reported
        Mock m; m.fail_at = k; auto o = exchange(m);
        CHECK(!o.ok && o.code == 5);
        CHECK(m.depth == 0);                         // the group is CLOSED
        CHECK(m.issued == k + 1);                    // nothing issued after the failure
        CHECK(m.log.back() == "end");
        const char* names[4] = {"send_down", "recv_below", "send_up", "recv_above"};
        CHECK(std::string(o.where) == names[k]);
    }
    {   // GroupStart fails: nothing opened, nothing issued, no GroupEnd
        Mock m; m.start_fails = 1; auto o = exchange(m);
        CHECK(!o.ok && o.code == 7 && m.depth == 0 && m.issued == 0 && m.log.empty());
        CHECK(std::string(o.where) == "ncclGroupStart");
    }
    {   // GroupEnd fails after clean operations: reported
        Mock m; m.end_fails = 1; auto o = exchange(m);
        CHECK(!o.ok && o.code == 9 && m.depth == 0 && std::string(o.where) == "ncclGroupEnd");
    }
    {   // an operation AND GroupEnd fail: the first failure wins
        Mock m; m.fail_at = 1; m.end_fails = 1; auto o = exchange(m);
        CHECK(!o.ok && o.code == 5 && std::string(o.where) == "recv_below" && m.depth == 0);
    }
    if (failures == 0) std::printf("rccl_group: all checks passed\n");
    return failures ? 1 : 0;
}
