// switches.h -- every environment switch of the library in ONE registry (switches.cpp).  The reference passes fixed flags to its engines
// (itsxpress/SeqSample.py:106-116,147-161,191-209) and has no such surface; this library's switches are performance knobs and
// diagnostics of its own making, so the rule is: nothing a user can leave in the environment by accident may change a result.
//   TUNING  result-neutral by construction and by test (block sizes, budgets, stream counts, A/B arms of a kernel)
//   MODE    selects a documented behaviour that the C ABI also offers as a call or an argument (rows mode, query masking)
//   DIAG    prints, counts or re-checks; results unchanged
//   HOOK    changes results or leaves work undone, for tests and experiments: honoured ONLY when ITSX_TEST_HOOKS=1, else ignored
// sw_get() is the only getenv of the library; itsx_switches() (include/itsx_hip.h) reports what is set, a search snapshots it.
#pragma once
#include <string>

namespace itsx {
enum SwKind { SW_TUNING = 0, SW_MODE = 1, SW_DIAG = 2, SW_HOOK = 3 };
struct Switch { const char *name; SwKind kind; const char *what; };
const Switch *sw_registry(int *n);
// the variable's value when it is set AND honoured (a HOOK needs ITSX_TEST_HOOKS=1), else nullptr
const char *sw_get(const char *name);
// "NAME=value" lines of every registered switch set in the environment right now (HOOKs that are not honoured say so)
std::string sw_report();
}  // namespace itsx
// (callers outside namespace itsx: the C ABI's functions, the host I/O code)
using itsx::sw_get;
