// TEST INFRASTRUCTURE -- DISCLOSED STAND-IN, the only one on the update path of oracle/_ref.
//
// The reference's cappedgd calls boost::math::quadrature::gauss<double, 15>::integrate (cnF2freq.cpp:4150).
// Boost.Math is not in this image (demo.sh:6 names boost_1_61_0; quadrature/gauss.hpp appeared in Boost 1.66),
// so this header supplies that one entry point with the algorithm Boost publishes for it
// (boost/math/quadrature/gauss.hpp, gauss<Real, N>::integrate(F, Real a, Real b) for finite a, b):
//   a == b -> 0;  b < a -> -integrate(f, b, a);
//   avg = (a + b) / 2, scale = (b - a) / 2, g(z) = f(avg + scale * z);
//   odd N: result = g(0) * w[0]; then for i = 1 .. (N - 1) / 2 in ascending abscissa:
//   result += (g(x[i]) + g(-x[i])) * w[i];  return result * scale.
// Node/weight tables: the 15-point Gauss-Legendre rule (Abramowitz & Stegun 25.4.30), the values Boost tabulates
// for double; tests/test_host_update.py checks them against numpy.polynomial.legendre.leggauss(15).
//
// Everything else on the update path of oracle/_ref (caplogitchange, cappedgd, processinfprobs, relskewhmm,
// updatehaploweights) is the reference's own text compiled from /root/reference.  Because of this stand-in the
// formal parity status of the update path stays "unpinned" (DESIGN.md section 2).
#pragma once
namespace boost { namespace math { namespace quadrature {

template<class Real, unsigned N> struct gauss;

template<> struct gauss<double, 15>
{
	template<class F> static double unit(F g)
	{
		static const double x[8] = { 0.00000000000000000e+00, 2.01194093997434522e-01, 3.94151347077563370e-01,
			5.70972172608538848e-01, 7.24417731360170047e-01, 8.48206583410427216e-01, 9.37273392400705904e-01,
			9.87992518020485428e-01 };
		static const double w[8] = { 2.02578241925561273e-01, 1.98431485327111576e-01, 1.86161000015562211e-01,
			1.66269205816993934e-01, 1.39570677926154314e-01, 1.07159220467171935e-01, 7.03660474881081247e-02,
			3.07532419961172684e-02 };
		double result = g(0.0) * w[0];
		for (unsigned i = 1; i < 8; ++i)
		{
			double fp = g(x[i]);
			double fm = g(-x[i]);
			result += (fp + fm) * w[i];
		}
		return result;
	}

	template<class F> static double integrate(F f, double a, double b)
	{
		if (a == b) return 0.0;
		if (b < a) return -integrate(f, b, a);
		double avg = (a + b) * 0.5;
		double scale = (b - a) * 0.5;
		double result = unit([&f, &avg, &scale](double z) -> double { return f(avg + scale * z); });
		result *= scale;
		return result;
	}
};

} } }
