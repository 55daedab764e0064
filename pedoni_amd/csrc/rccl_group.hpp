// rccl_group.hpp -- the control flow of one grouped RCCL exchange, free of HIP / RCCL types so
// that tests/cpp/test_rccl_group.cpp can run it on the CPU against a mock.
//
// ncclGroupStart ... ncclGroupEnd brackets the sends and receives a rank issues to its neighbours.
// Whatever fails in between, the group MUST be closed before the caller returns: a rank that leaves
// a group open keeps every later call of its thread inside that group (nothing is ever launched)
// while its peers sit in theirs, waiting.  So: after the first failure no further operation is
// issued, GroupEnd is still called, and the FIRST failure is the one reported.
#pragma once

template <typename Result> struct GroupOutcome {
    bool ok;
    Result code;        // first failing call's result
    const char* where;  // and its name
};

// start() / end(): the bracket; body(op) issues the operations as op(call, name) where call()
// returns a Result.  `success` is the Result that means "no error".
template <typename Result, typename Start, typename End, typename Body>
GroupOutcome<Result> run_group(Result success, Start&& start, End&& end, Body&& body)
{
    GroupOutcome<Result> out{true, success, ""};
    const Result s = start();
    if (s != success) return GroupOutcome<Result>{false, s, "ncclGroupStart"};   // nothing was opened
    auto op = [&](auto&& call, const char* name) {
        if (!out.ok) return;                       // after a failure: issue nothing more
        const Result r = call();
        if (r != success) out = GroupOutcome<Result>{false, r, name};
    };
    body(op);
    const Result e = end();                        // always: the group is closed on every path
    if (out.ok && e != success) out = GroupOutcome<Result>{false, e, "ncclGroupEnd"};
    return out;
}
