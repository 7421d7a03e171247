// dwt_lift.h -- lifting arithmetic shared by every HIP kernel of the path.
//
// One policy struct per wavelet/type.  A policy states the lifting steps exactly as
// the reference evaluates them, so that results are bit-identical to libdwt's CPU
// path (built without FMA: arch.mk:15,38-39 -- this file must be compiled with
// -ffp-contract=off):
//
//   CDF 9/7 float  src/libdwt.c:2264-2355 (accel_lift_op4s_main_s), constants
//                  src/inline.h:309-315; forward call :10780, inverse call :11561
//   CDF 5/3 int32  src/libdwt.c:10950-10984 / 11749-11783
//   CDF 5/3 float  src/libdwt.c:10986-11030 / 11785-11829, constants inline.h:331-335
//
// Conventions.  A line is an interleaved signal a[0..N): even samples become L
// (s), odd samples become H (d).  Ends use whole-sample symmetric reflection: at index 0 and
// N-1 both taps of a sample are one and the same sample x.  The reference does not evaluate
// its interior formula there but an END FORM of its own: the float and double kernels add
// `2*c*x`, i.e. (2c)*x (:9545-9552, :9873-9907, :10994-11017, :2024-2083) -- the same bits as
// c*(x+x) unless x+x overflows (|x| > FLT_MAX/2) --, the int 5/3 `(d+1)>>1` and `-= s`
// (:10971-10976, :11768-11773), which differ from the reflected `(d+d+2)>>2` / `(s+s)>>1` once
// the doubled term wraps (|x| >= 2^30); only the fixed-point 9/7 writes `a[N-2]+a[N-2]` itself.
// So the policies carry explicit END FORMS (kEndForms, fwd_end / inv_end) and the kernels
// apply them to the samples whose two taps are one and the same sample.  The tile sweeps do it by SELECTION at the two
// window entries that can be a line end (SelEnds below; float: the end step is the plain step with the coefficient doubled
// and the virtual tap replaced by -0.0 -- straight-line code, no second formula; int 5/3: both formulas, one select); the
// shapes SelEnds leaves out take a path of their own in the waves whose tile holds a line end (wave-uniform tests).
// Int arithmetic wraps modulo 2^32 like the compiled reference
// (done in unsigned here: signed overflow is undefined for the compiler).  A forward transform runs K lifting
// steps, step s acting on samples of parity (s+1)&1, then scales; an inverse
// transform descales, then runs K steps, step s acting on parity s&1.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

// The float / double policies' line-end forms are a BUILD option, for A/B timing.  1 (libdwt_hip.so, the default): every 2-D
// kernel applies the reference's own form, bit for bit over the whole float range (tests/test_hip_float_range.py).  0
// (`make plain` -> libdwt_hip_plain.so): the line ends are what reflection gives, c*(x+x) -- the same bits unless x+x
// overflows; what shipped until the select form made the exact ends free (round 6, alternated on one box: forward call of
// one 8192^2 image 155 = 155 us, the bench's batch of 64 unchanged within its noise, inverse calls + 2 % before the
// packed two-row lift gave that back; the branching forms had cost 8-11 % and 3.5 %: profiles/r06_notes.md).  The 3-D level kernels keep the reflected form in
// either build.  The int 5/3 has its end forms in every build.
#ifndef DWT_FLOAT_END_FORMS
#define DWT_FLOAT_END_FORMS 1
#endif

namespace dwt {

struct Cdf97S {
	using T = float;
	static constexpr bool kEndForms = DWT_FLOAT_END_FORMS != 0; // the reference adds (2c)*x at a line end
	static constexpr int K = 4;          // lifting steps; also the halo in samples
	static constexpr bool kScaleSingle = true;   // N==1 lines are scaled (:10757, :11546)
	static constexpr bool kSkipSingleLine = true; // 2-D drivers skip a direction with one line (:12837)
	static constexpr bool kInvColsFirst = false;  // inverse: rows then columns (:17098-17154)
	// -p1, u1, -p2, u2 as the forward call passes them (:10780)
	static __device__ __forceinline__ T fc(int s)
	{
		return s == 0 ? -1.58613434342059f : s == 1 ? -0.0529801185729f : s == 2 ? 0.8829110755309f : 0.4435068520439f;
	}
	// -u2, p2, -u1, p1 as the inverse call passes them (:11561)
	static __device__ __forceinline__ T ic(int s)
	{
		return s == 0 ? -0.4435068520439f : s == 1 ? -0.8829110755309f : s == 2 ? 0.0529801185729f : 1.58613434342059f;
	}
	static __device__ __forceinline__ T zeta() { return 1.1496043988602f; }
	static __device__ __forceinline__ T fwd_step(int s, T c, T l, T r) { return c + fc(s) * (l + r); }
	static __device__ __forceinline__ T inv_step(int s, T c, T l, T r) { return c + ic(s) * (l + r); }
	// forward: even *= zeta, odd *= 1/zeta with 1/zeta a float division (:2327, :2344-2353)
	static __device__ __forceinline__ T fwd_scale(int parity, T v) { return parity ? v * (1.0f / zeta()) : v * zeta(); }
	// inverse: even *= 1/zeta, odd *= zeta (:2292-2300)
	static __device__ __forceinline__ T inv_scale(int parity, T v) { return parity ? v * zeta() : v * (1.0f / zeta()); }
	// N==1: forward *s1 (:10759); inverse *s2 where s2 = (float)(1/1.1496043988602) (inline.h:315)
	static __device__ __forceinline__ T fwd_single(T v) { return v * zeta(); }
	static __device__ __forceinline__ T inv_single(T v) { return v * (float)(1 / 1.1496043988602); }
	// line ends: both taps are the sample m; `2*alpha*(x)` as the reference writes it (:9545, :9552, :9873, :9879)
	static __device__ __forceinline__ T fwd_end(int s, T c, T m) { return c + (2.0f * fc(s)) * m; }
	static __device__ __forceinline__ T inv_end(int s, T c, T m) { return c + (2.0f * ic(s)) * m; }
	// the steps as c + k (l + r) with k a value (SelEnds below): fk / ik = the signed coefficient of step s
	static __device__ __forceinline__ T fk(int s) { return fc(s); }
	static __device__ __forceinline__ T ik(int s) { return ic(s); }
	static __device__ __forceinline__ T step_k(T k, T c, T l, T r) { return c + k * (l + r); }
};

// Same wavelet with each lifting step contracted to one fused multiply-add.  NOT the
// reference's rounding (it has no FMA): results differ in the last bits, well inside the
// 1e-5 relative tolerance of the north star.  Opt-in only (option "fma").
struct Cdf97SFma : Cdf97S {
	static __device__ __forceinline__ T fwd_step(int s, T c, T l, T r) { return __builtin_fmaf(fc(s), l + r, c); }
	static __device__ __forceinline__ T inv_step(int s, T c, T l, T r) { return __builtin_fmaf(ic(s), l + r, c); }
	static __device__ __forceinline__ T fwd_end(int s, T c, T m) { return __builtin_fmaf(2.0f * fc(s), m, c); }
	static __device__ __forceinline__ T inv_end(int s, T c, T m) { return __builtin_fmaf(2.0f * ic(s), m, c); }
	static __device__ __forceinline__ T step_k(T k, T c, T l, T r) { return __builtin_fmaf(k, l + r, c); }
};

struct Cdf53I {
	using T = int;
	static constexpr int K = 2;
#ifdef DWT_DEBUG_NO_INT_END_FORMS
	static constexpr bool kEndForms = false;
#else
	static constexpr bool kEndForms = true;       // :10971-10976, :11768-11773
#endif
	static constexpr bool kScaleSingle = false;   // N<2 untouched (:10961)
	static constexpr bool kSkipSingleLine = false;
	static constexpr bool kInvColsFirst = true;   // inverse: columns then rows (:18178-18195)
	// wrapping int32 sums (the compiled reference wraps; the shifts are arithmetic)
	static __device__ __forceinline__ T add(T a, T b) { return (T)((unsigned)a + (unsigned)b); }
	static __device__ __forceinline__ T sub(T a, T b) { return (T)((unsigned)a - (unsigned)b); }
	static __device__ __forceinline__ T fwd_step(int s, T c, T l, T r)
	{
		return s == 0 ? sub(c, add(l, r) >> 1) : add(c, add(add(l, r), 2) >> 2);
	}
	static __device__ __forceinline__ T inv_step(int s, T c, T l, T r)
	{
		return s == 0 ? sub(c, add(add(l, r), 2) >> 2) : add(c, add(l, r) >> 1);
	}
	// line ends: both taps are the sample m.  Forward (:10971-10976): the last odd sample of an
	// even-length line `-= s`, sample 0 and the last even sample of an odd-length line
	// `+= (d+1)>>1`; inverse (:11768-11773) the same terms with the opposite sign.
	static __device__ __forceinline__ T fwd_end(int s, T c, T m) { return s == 0 ? sub(c, m) : add(c, add(m, 1) >> 1); }
	static __device__ __forceinline__ T inv_end(int s, T c, T m) { return s == 0 ? sub(c, add(m, 1) >> 1) : add(c, m); }
	static __device__ __forceinline__ T fwd_scale(int, T v) { return v; }
	static __device__ __forceinline__ T inv_scale(int, T v) { return v; }
	static __device__ __forceinline__ T fwd_single(T v) { return v; }
	static __device__ __forceinline__ T inv_single(T v) { return v; }
};

// Fixed-point int32 CDF 9/7 (src/libdwt.c:10901-10948, 11699-11746): no scaling; the
// reference's own end formulas are the reflected ones (`a[N-2]+a[N-2]`), so reflection
// is exact here for any input.
struct Cdf97I {
	using T = int;
	static constexpr bool kEndForms = false; // the reflected taps ARE the reference's end formulas
	static constexpr int K = 4;
	static constexpr bool kScaleSingle = false;
	static constexpr bool kSkipSingleLine = false;
	static constexpr bool kInvColsFirst = true; // :18256-18274: columns, then rows
	// wrapping int32 arithmetic (as the compiled reference behaves; unsigned for the compiler)
	static __device__ __forceinline__ T add(T a, T b) { return (T)((unsigned)a + (unsigned)b); }
	static __device__ __forceinline__ T sub(T a, T b) { return (T)((unsigned)a - (unsigned)b); }
	// (k * (l + r) + bias) >> sh
	static __device__ __forceinline__ T term(int k, T l, T r, int bias, int sh)
	{
		return (T)((unsigned)k * ((unsigned)l + (unsigned)r) + (unsigned)bias) >> sh;
	}
	static __device__ __forceinline__ T fwd_step(int s, T c, T l, T r)
	{
		return s == 0 ? sub(c, term(+203, l, r, -(1 << 6), 7))
		     : s == 1 ? add(c, term(-217, l, r, 1 << 11, 12))
		     : s == 2 ? sub(c, term(-113, l, r, -(1 << 6), 7))
		              : add(c, term(1817, l, r, 1 << 11, 12));
	}
	static __device__ __forceinline__ T inv_step(int s, T c, T l, T r)
	{
		return s == 0 ? sub(c, term(1817, l, r, 1 << 11, 12))
		     : s == 1 ? add(c, term(-113, l, r, -(1 << 6), 7))
		     : s == 2 ? sub(c, term(-217, l, r, 1 << 11, 12))
		              : add(c, term(+203, l, r, -(1 << 6), 7));
	}
	static __device__ __forceinline__ T fwd_scale(int, T v) { return v; }
	static __device__ __forceinline__ T inv_scale(int, T v) { return v; }
	static __device__ __forceinline__ T fwd_single(T v) { return v; }
	static __device__ __forceinline__ T inv_single(T v) { return v; }
};

// The interleaved in-place int 9/7 drivers (src/libdwt.c:17356-17422 forward, :17237-17306
// inverse) ADD the rounded terms where the Mallat int kernels subtract them:
// (-203*s + 64) >> 7 is not -((203*s - 64) >> 7) in floor arithmetic.
struct Cdf97IIp : Cdf97I {
	static __device__ __forceinline__ T fwd_step(int s, T c, T l, T r)
	{
		return s == 0 ? add(c, term(-203, l, r, 1 << 6, 7))
		     : s == 1 ? add(c, term(-217, l, r, 1 << 11, 12))
		     : s == 2 ? add(c, term(+113, l, r, 1 << 6, 7))
		              : add(c, term(1817, l, r, 1 << 11, 12));
	}
	static __device__ __forceinline__ T inv_step(int s, T c, T l, T r)
	{
		return s == 0 ? sub(c, term(1817, l, r, 1 << 11, 12))
		     : s == 1 ? sub(c, term(+113, l, r, 1 << 6, 7))
		     : s == 2 ? sub(c, term(-217, l, r, 1 << 11, 12))
		              : sub(c, term(-203, l, r, 1 << 6, 7));
	}
};

struct Cdf53S {
	using T = float;
	static constexpr bool kEndForms = DWT_FLOAT_END_FORMS != 0; // the reference adds (2c)*x at a line end
	static constexpr int K = 2;
	static constexpr bool kScaleSingle = true;    // :10998-11003, :11797-11802
	static constexpr bool kSkipSingleLine = false; // :16507-16523 run unconditionally
	static constexpr bool kInvColsFirst = false;  // :18333-18349
	static __device__ __forceinline__ T s1() { return 1.41421356237309504880f; }
	static __device__ __forceinline__ T s2() { return 0.70710678118654752440f; }
	static __device__ __forceinline__ T fwd_step(int s, T c, T l, T r)
	{
		return s == 0 ? c - 0.5f * (l + r) : c + 0.25f * (l + r);
	}
	static __device__ __forceinline__ T inv_step(int s, T c, T l, T r)
	{
		return s == 0 ? c - 0.25f * (l + r) : c + 0.5f * (l + r);
	}
	static __device__ __forceinline__ T fwd_scale(int parity, T v) { return parity ? v * s2() : v * s1(); }
	static __device__ __forceinline__ T inv_scale(int parity, T v) { return parity ? v * s1() : v * s2(); }
	static __device__ __forceinline__ T fwd_single(T v) { return v * s1(); }
	static __device__ __forceinline__ T inv_single(T v) { return v * s2(); }
	// line ends (:11012-11017, :11811-11816): `-= 2*p1*x`, `+= 2*u1*x`
	static __device__ __forceinline__ T fwd_end(int s, T c, T m) { return s == 0 ? c - (2 * 0.5f) * m : c + (2 * 0.25f) * m; }
	static __device__ __forceinline__ T inv_end(int s, T c, T m) { return s == 0 ? c - (2 * 0.25f) * m : c + (2 * 0.5f) * m; }
	// (c - p t and c + (-p) t are the same float for every c, t)
	static __device__ __forceinline__ T fk(int s) { return s == 0 ? -0.5f : 0.25f; }
	static __device__ __forceinline__ T ik(int s) { return s == 0 ? -0.25f : 0.5f; }
	static __device__ __forceinline__ T step_k(T k, T c, T l, T r) { return c + k * (l + r); }
};

// dwt-simple.c's 5/3 (fdwt2_cdf53_*, :1031-1078, :1531-1570): same steps, but the odd
// coefficients are scaled by `1/zeta` computed in float instead of the stored s2.
struct Cdf53SNew : Cdf53S {
	static __device__ __forceinline__ T fwd_scale(int parity, T v) { return parity ? v * (1.0f / s1()) : v * s1(); }
};

// Double precision (src/libdwt.c:2024-2083, 11423-11482; constants src/inline.h:317-323).
// The reference writes the steps as `a -= p*(l+r)` / `a += u*(l+r)` and scales by the
// two stored constants s1, s2 = 1/1.1496043988602.
struct Cdf97D {
	using T = double;
	static constexpr bool kEndForms = DWT_FLOAT_END_FORMS != 0; // the reference adds (2c)*x at a line end
	static constexpr int K = 4;
	static constexpr bool kScaleSingle = true;
	static constexpr bool kSkipSingleLine = false; // :12490-12506 run unconditionally
	static constexpr bool kInvColsFirst = false;
	static __device__ __forceinline__ T p1() { return 1.58613434342059; }
	static __device__ __forceinline__ T u1() { return -0.0529801185729; }
	static __device__ __forceinline__ T p2() { return -0.8829110755309; }
	static __device__ __forceinline__ T u2() { return 0.4435068520439; }
	static __device__ __forceinline__ T s1() { return 1.1496043988602; }
	static __device__ __forceinline__ T s2() { return 1 / 1.1496043988602; }
	static __device__ __forceinline__ T fwd_step(int s, T c, T l, T r)
	{
		return s == 0 ? c - p1() * (l + r) : s == 1 ? c + u1() * (l + r) : s == 2 ? c - p2() * (l + r) : c + u2() * (l + r);
	}
	static __device__ __forceinline__ T inv_step(int s, T c, T l, T r)
	{
		return s == 0 ? c - u2() * (l + r) : s == 1 ? c + p2() * (l + r) : s == 2 ? c - u1() * (l + r) : c + p1() * (l + r);
	}
	static __device__ __forceinline__ T fwd_scale(int parity, T v) { return parity ? v * s2() : v * s1(); }
	static __device__ __forceinline__ T inv_scale(int parity, T v) { return parity ? v * s1() : v * s2(); }
	static __device__ __forceinline__ T fwd_single(T v) { return v * s1(); }
	static __device__ __forceinline__ T inv_single(T v) { return v * s2(); }
	// line ends (:2040-2047, :11439-11446): `+= 2*u*x`, `-= 2*p*x`
	static __device__ __forceinline__ T fwd_end(int s, T c, T m) { return s == 0 ? c - (2 * p1()) * m : s == 1 ? c + (2 * u1()) * m : s == 2 ? c - (2 * p2()) * m : c + (2 * u2()) * m; }
	static __device__ __forceinline__ T inv_end(int s, T c, T m) { return s == 0 ? c - (2 * u2()) * m : s == 1 ? c + (2 * p2()) * m : s == 2 ? c - (2 * u1()) * m : c + (2 * p1()) * m; }
	// the steps as c + k (l + r) (SelEnds; c - p t and c + (-p) t are the same double for every c, t)
	static __device__ __forceinline__ T fk(int s) { return s == 0 ? -p1() : s == 1 ? u1() : s == 2 ? -p2() : u2(); }
	static __device__ __forceinline__ T ik(int s) { return s == 0 ? -u2() : s == 1 ? p2() : s == 2 ? -u1() : p1(); }
	static __device__ __forceinline__ T step_k(T k, T c, T l, T r) { return c + k * (l + r); }
};

// src/libdwt.c:2085-2130, 11484-11530; constants src/inline.h:337-341
struct Cdf53D {
	using T = double;
	static constexpr bool kEndForms = DWT_FLOAT_END_FORMS != 0; // the reference adds (2c)*x at a line end
	static constexpr int K = 2;
	static constexpr bool kScaleSingle = true;
	static constexpr bool kSkipSingleLine = false;
	static constexpr bool kInvColsFirst = false;
	static __device__ __forceinline__ T s1() { return 1.41421356237309504880; }
	static __device__ __forceinline__ T s2() { return 0.70710678118654752440; }
	static __device__ __forceinline__ T fwd_step(int s, T c, T l, T r) { return s == 0 ? c - 0.5 * (l + r) : c + 0.25 * (l + r); }
	static __device__ __forceinline__ T inv_step(int s, T c, T l, T r) { return s == 0 ? c - 0.25 * (l + r) : c + 0.5 * (l + r); }
	static __device__ __forceinline__ T fwd_scale(int parity, T v) { return parity ? v * s2() : v * s1(); }
	static __device__ __forceinline__ T inv_scale(int parity, T v) { return parity ? v * s1() : v * s2(); }
	static __device__ __forceinline__ T fwd_single(T v) { return v * s1(); }
	static __device__ __forceinline__ T inv_single(T v) { return v * s2(); }
	static __device__ __forceinline__ T fwd_end(int s, T c, T m) { return s == 0 ? c - (2 * 0.5) * m : c + (2 * 0.25) * m; }
	static __device__ __forceinline__ T inv_end(int s, T c, T m) { return s == 0 ? c - (2 * 0.25) * m : c + (2 * 0.5) * m; }
	static __device__ __forceinline__ T fk(int s) { return s == 0 ? -0.5 : 0.25; }
	static __device__ __forceinline__ T ik(int s) { return s == 0 ? -0.25 : 0.5; }
	static __device__ __forceinline__ T step_k(T k, T c, T l, T r) { return c + k * (l + r); }
};

// Whole-sample symmetric reflection of i into [0, N), N >= 2, any i.
static __device__ __forceinline__ int reflect(int i, int N)
{
	const int period = 2 * (N - 1);
	i %= period;
	if (i < 0)
		i += period;
	return i < N ? i : period - i;
}

// Single-bounce reflection, valid for -N < i < 2N-1 (the only range a halo of K
// samples needs when N > K).
static __device__ __forceinline__ int reflect1(int i, int N)
{
	i = i < 0 ? -i : i;
	return i >= N ? 2 * (N - 1) - i : i;
}

// W with its line ends as reflection gives them (FwdLevelArgs::plain_ends: the xy sweeps of the 3-D transforms)
template <class W>
struct PlainEnds : W {
	static constexpr bool kEndForms = false;
};

// reflect() without the division, for an index within a few samples of a line of 16 samples or more -- one bounce; whatever
// lies further out (the columns of lanes beyond the line's end, whose results are dropped) is clamped into the line so
// that it stays a valid address
static __device__ __forceinline__ int reflect_near(int i, int N)
{
	i = i < 0 ? -i : i;
	i = i >= N ? 2 * (N - 1) - i : i;
	return i < 0 ? 0 : i;
}

// W with its line-end forms by SELECTION instead of by branches (float policies with fk / ik / step_k): at a line end the
// reference adds (2c) x where both taps of the plain step are that x.  c + k (l + r) gives those very bits with k = 2c and
// the virtual tap replaced by -0.0 (x + -0.0 == x for every x, +-0, Inf and NaN included) -- a select on the coefficient
// and one on a tap instead of a second formula behind a branch.  The tile sweeps run this instantiation on every tile
// of a level whose width is a multiple of the columns per lane: straight-line code, the same for all waves (the
// branching forms cost the sweeps 4-10 %, most of it code layout and scheduling, not arithmetic: profiles/r06_notes.md).
template <class W>
struct SelEnds : W {
	static constexpr bool kEndForms = false;
};
template <class W> constexpr bool kIsSelEnds = false;
template <class W> constexpr bool kIsSelEnds<SelEnds<W>> = true;
template <class W, class = void> struct has_coef_ends : std::false_type {};
template <class W> struct has_coef_ends<W, std::void_t<decltype(W::fk(0))>> : std::true_type {};

// One step of the select form on a sample that can be a line end (`e`): `virt` is the tap that is virtual there, `real` the
// other (the same values at an end).  Float policies: the plain step with the lane's coefficient kk (doubled where e) and
// the virtual tap dropped; the int 5/3, whose end forms are formulas of their own, evaluates both and selects.
template <class W, bool INV>
static __device__ __forceinline__ typename W::T sel_step(int s, bool e, typename W::T kk, typename W::T c, typename W::T virt, typename W::T real)
{
	using T = typename W::T;
	if constexpr (has_coef_ends<W>::value)
		return W::step_k(kk, c, e ? T(-0.0) : virt, real);
	else
		return e ? (INV ? W::inv_end(s, c, real) : W::fwd_end(s, c, real)) : (INV ? W::inv_step(s, c, virt, real) : W::fwd_step(s, c, virt, real));
}
// the coefficient of step s for sel_step: doubled where the sample is an end (float policies; nothing for the int 5/3)
template <class W, bool INV>
static __device__ __forceinline__ typename W::T sel_coef(int s, bool e)
{
	using T = typename W::T;
	if constexpr (has_coef_ends<W>::value) {
		const T k = INV ? W::ik(s) : W::fk(s);
		return e ? T(2) * k : k;
	} else
		return T(0);
}

// The K lifting steps over a[0..n) as lift_fwd_regs / lift_inv_regs run them, with two entries that can be line ends:
// J0 (column 0: its LEFT tap is virtual) when e0, J1 (the last column: its RIGHT tap is virtual) when e1.  kk[s]: the
// lane's coefficient of step s for those entries (doubled where the entry the step reaches is an end).
template <class W, int n, bool INV, int J0, int J1>
static __device__ __forceinline__ void lift_regs_sel(typename W::T (&a)[n], bool e0, bool e1, const typename W::T (&kk)[W::K])
{
	static_assert(((J0 ^ J1) & 1) == 1, "the two candidates are reached by different steps");
#pragma unroll
	for (int s = 0; s < W::K; s++) {
#pragma unroll
		for (int j = s + 1; j <= n - 2 - s; j += 2) {
			if (j == J0)
				a[j] = sel_step<W, INV>(s, e0, kk[s], a[j], a[j - 1], a[j + 1]);
			else if (j == J1)
				a[j] = sel_step<W, INV>(s, e1, kk[s], a[j], a[j + 1], a[j - 1]);
			else
				a[j] = INV ? W::inv_step(s, a[j], a[j - 1], a[j + 1]) : W::fwd_step(s, a[j], a[j - 1], a[j + 1]);
		}
	}
}

// the lane's coefficients for lift_regs_sel
template <class W, bool INV, int J0>
static __device__ __forceinline__ void sel_coefs(typename W::T (&kk)[W::K], bool e0, bool e1)
{
#pragma unroll
	for (int s = 0; s < W::K; s++)
		kk[s] = sel_coef<W, INV>(s, ((J0 - (s + 1)) & 1) == 0 ? e0 : e1); // (the candidate step s reaches)
}

// Step s on a sample whose taps are l and r; `end`: the sample sits on a line end (index 0 or
// N-1 after reflection), where l and r are one and the same sample.  Only policies with explicit
// end forms look at `end`.
template <class W>
static __device__ __forceinline__ typename W::T fwd_step_at(int s, bool end, typename W::T c, typename W::T l, typename W::T r)
{
	if constexpr (W::kEndForms)
		return end ? W::fwd_end(s, c, l) : W::fwd_step(s, c, l, r);
	else
		return W::fwd_step(s, c, l, r);
}

template <class W>
static __device__ __forceinline__ typename W::T inv_step_at(int s, bool end, typename W::T c, typename W::T l, typename W::T r)
{
	if constexpr (W::kEndForms)
		return end ? W::inv_end(s, c, l) : W::inv_step(s, c, l, r);
	else
		return W::inv_step(s, c, l, r);
}

// Bit j of the result: sample g0 + j of a line of N samples (N >= 2, any g0: reflected) is a line end.
template <int n>
static __device__ __forceinline__ unsigned end_mask(int g0, int N)
{
	// A window that leaves [0, N) by less than N on either side (every window of a tile sweep over a line of 64 samples or
	// more): a sample is an end iff it IS sample 0 or N - 1 -- one bounce maps no other index onto them.  Two shifts instead
	// of n reflections with their integer divisions (which cost the small, latency-bound levels of one image 2-3 us each).
	if (N >= 64 && g0 > -N && g0 + n <= 2 * N - 1) {
		const int j0 = -g0, j1 = N - 1 - g0;
		return ((unsigned)j0 < (unsigned)n ? 1u << j0 : 0u) | ((unsigned)j1 < (unsigned)n ? 1u << j1 : 0u);
	}
	unsigned m = 0;
#pragma unroll
	for (int j = 0; j < n; j++) {
		const int i = reflect(g0 + j, N);
		m |= (unsigned)(i == 0 || i == N - 1) << j;
	}
	return m;
}

// end_mask for a window of a tile sweep over a line of 64 samples or more that starts at g0 > -N: the one-bounce form
// alone (windows that reach beyond 2N - 1 belong to lanes right of the line, whose results are dropped)
template <int n>
static __device__ __forceinline__ unsigned end_mask_long(int g0, int N)
{
	const int j0 = -g0, j1 = N - 1 - g0;
	return ((unsigned)j0 < (unsigned)n ? 1u << j0 : 0u) | ((unsigned)j1 < (unsigned)n ? 1u << j1 : 0u);
}

// Run the K lifting steps of a forward transform over a register array a[0..n)
// whose element 0 is an EVEN sample.  After step s, entries j in [s+1, n-2-s] of
// parity (s+1)&1 are valid; the caller takes the centre it needs.  Fully unrolled:
// all indices are compile-time constants.
// `ends`: end_mask of the array (used by policies with explicit end forms only).  CAND: the entries that CAN be a line
// end for this caller (compile time): only they test their bit of `ends`, the others take the plain step -- a tile sweep
// whose tiles start at multiples of its width meets column 0 and the last column at two fixed entries, so that its
// border tiles pay two tests per step pair instead of one per entry (measured: the full mask cost the sweeps 10 %).
template <class W, int n, unsigned CAND = ~0u>
static __device__ __forceinline__ void lift_fwd_regs(typename W::T (&a)[n], unsigned ends = 0)
{
#pragma unroll
	for (int s = 0; s < W::K; s++) {
#pragma unroll
		for (int j = s + 1; j <= n - 2 - s; j += 2)
			a[j] = fwd_step_at<W>(s, ((CAND >> j) & 1) && ((ends >> j) & 1), a[j], a[j - 1], a[j + 1]);
	}
}

// Inverse steps over a[0..n) whose element 0 is an ODD sample (so step 0, which
// acts on even samples, again starts at j = 1).  Entries must be descaled first.
template <class W, int n, unsigned CAND = ~0u>
static __device__ __forceinline__ void lift_inv_regs(typename W::T (&a)[n], unsigned ends = 0)
{
#pragma unroll
	for (int s = 0; s < W::K; s++) {
#pragma unroll
		for (int j = s + 1; j <= n - 2 - s; j += 2)
			a[j] = inv_step_at<W>(s, ((CAND >> j) & 1) && ((ends >> j) & 1), a[j], a[j - 1], a[j + 1]);
	}
}

} // namespace dwt
