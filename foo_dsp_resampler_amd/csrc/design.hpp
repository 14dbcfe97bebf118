// Host-side (fp64) filter design for the MI355X SoX-rate engine.
//
// Runs once per RR_open on the host, as the reference does (rate/effects_i_dsp.c); nothing here
// touches the GPU.  Function-level parity notes cite /root/reference/ file:line.
#pragma once
#include <complex>
#include <vector>

namespace rsmp {

using cplx = std::complex<double>;

// In-place radix-2 complex FFT, Y[k] = sum_j a[j] exp(sign * 2 pi i j k / n); n = power of two.
void fft_inplace(std::vector<cplx> &a, int sign);

// Kaiser-windowed-sinc low-pass design; behaviour of lsx_design_lpf with CREATE_4X_NUMTAPS
// (rate/effects_i_dsp.c:137-171, rate/sox_i.h:17).
//   k > 0 : polyphase prototype with k phases (taps-per-phase rounded up to a multiple of 4)
//   k < 0 : num_taps == 1 (mod -k)
//   Fn < 0: sizing run only (returns an empty vector, num_taps still set)
std::vector<double> design_lowpass(double Fp, double Fs, double Fn, double att_dB, int &num_taps, int k,
                                   double beta = -1.0);

// Linear -> minimum/intermediate phase conversion (rate/effects_i_dsp.c:181-278).
// phase in [0,100]; 50 is never passed (linear phase keeps the symmetric design as is).
// On return h may have a new length; post_len = number of taps after the impulse peak.
void to_phase(std::vector<double> &h, int &post_len, double phase);

// DFT block length for an overlap-save stage (rate/effects_i_dsp.c:64-73).
int dft_block_length(int num_taps);

double bessel_i0(double x);                     // rate/effects_i_dsp.c:46-55
double kaiser_beta(double att_dB, double tr_bw); // rate/effects_i_dsp.c:83-108

} // namespace rsmp
