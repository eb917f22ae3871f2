// Every environment variable the library (and its Python wrapper) looks at, in ONE table: name, class, what it does.  No other file
// calls getenv() -- tests/test_host_logic.py greps for that -- and kv_knob() answers for registered names only.
//
//   SETTING     a user may set it; always honoured (cache sizes, host threads, which ingest path reads a file, progress on stderr).
//   TUNING      pins a path or shrinks a geometry so that tests and the A/B runners under scratch/ reach code that real inputs reach
//               only at full size.  RESULTS ARE THE SAME on every path (the parity tests hold each against the oracle).  Honoured
//               only while KV_TUNING=1 is set: a stray KV_* in a user's shell changes nothing.
//   EXPERIMENT  timing dissection that skips parts of kernels: RESULTS ARE WRONG.  Honoured only by a library built with
//               -DKV_EXPERIMENTS (scratch/ab_build.py does; __graft_entry__.build_product never) and KV_TUNING=1.
//
// kv_knobs_describe() (include/kvsketch.h) lists what is set and honoured right now -- bench.py prints it in its JSON line -- or the
// whole table.
#pragma once
#include <stddef.h>

enum KvKnobClass { KV_KNOB_SETTING = 0, KV_KNOB_TUNING = 1, KV_KNOB_EXPERIMENT = 2 };
struct KvKnobDef {
    const char *name;
    KvKnobClass cls;
    const char *doc;
};

// value of a registered knob if it is set AND its class is honoured right now, else nullptr.  An unregistered name is a programming
// error: nullptr, and one line on stderr.
const char *kv_knob(const char *name);
const KvKnobDef *kv_knob_table(size_t *n);
