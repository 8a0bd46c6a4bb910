// alias.hxx -- the `sparsify::` spelling of the namespace (BASELINE north_star: sparsify::sparsify, sparsify::spmma, sparsify::spmm,
// sparsify::gemm), OPT-IN: compile with -DSPARSIFYME_NAMESPACE_ALIAS and every header of this directory also answers to
//   sparsify::sparsify<2, 2>(w, mask, m, n);   sparsify::spmma(dA, dB, dC, m, n, k, b);
//   sparsify::batched::gemm(...);              sparsify::batched::spmm(...);
// The reference's namespace is `sparsifyme` (include/sparsify.me/sparsify.hxx:19) and its drivers say `using namespace sparsifyme;`
// (examples/spmma.cu:23, spmm.cu:25): next to that directive a GLOBAL alias named `sparsify` would make the unqualified call
// `sparsify<2, 2>(...)` of examples/sparsify.cu:46 ambiguous (function template vs namespace) -- hence a macro, off by default, and
// never together with the using-directive in one translation unit.
#pragma once
#ifdef SPARSIFYME_NAMESPACE_ALIAS
namespace sparsifyme {}
namespace sparsify = sparsifyme;
#endif
