//! `fourier` plans on an MI355X: the reference's plan API over the HIP engine in `libfourier.so`.
//!
//! This is `fourier-ffi` reversed.  There (`fourier-ffi/src/lib.rs:3-106`) C calls Rust plans through
//! `fourier_create_* / fourier_transform_*`; here Rust calls the same eight symbols, now exported by the HIP
//! engine (`include/fourier.h` Part 1), plus the device-resident batched extension (Part 2).  The public
//! surface is the reference's own: [`create_fft_f32`] / [`create_fft_f64`] with the signatures of
//! `fourier/src/lib.rs:31,49`, returning `Box<dyn Fft<Real = T> + Send>`; [`Fft`] and [`Transform`] are
//! re-exported from `fourier-algorithms` unchanged (`fourier-algorithms/src/fft.rs:4-82`).
//!
//! Call sequence the shim relies on (exercised from C by `tests/c/ffi_sequence.c` in this repository, because
//! this crate cannot be compiled in the build image): `create` -> `transform_in_place` -> `transform` ->
//! `destroy`, for both precisions; `create` returns NULL on failure; every other legacy call is `void` and
//! never unwinds; an unknown transform code is a no-op.
//!
//! Plans are `Send`, not `Sync`, like the reference's (`RefCell` scratch, `autosort/mod.rs:54`): the engine
//! keeps per-handle scratch and a private stream.

pub use fourier_algorithms::{Fft, Transform};

#[cfg(feature = "hip")]
mod hip {
    use super::{Fft, Transform};
    use num_complex::Complex;
    use std::os::raw::{c_char, c_int, c_void};

    #[repr(C)]
    pub struct FourierFftFloat {
        _private: [u8; 0],
    }
    #[repr(C)]
    pub struct FourierFftDouble {
        _private: [u8; 0],
    }

    extern "C" {
        // include/fourier.h Part 1 == fourier-ffi/include/fourier.h:41-58
        fn fourier_create_float(size: usize) -> *mut FourierFftFloat;
        fn fourier_destroy_float(p: *mut FourierFftFloat);
        fn fourier_transform_in_place_float(p: *const FourierFftFloat, x: *mut Complex<f32>, t: c_int);
        fn fourier_transform_float(p: *const FourierFftFloat, i: *const Complex<f32>, o: *mut Complex<f32>, t: c_int);
        fn fourier_create_double(size: usize) -> *mut FourierFftDouble;
        fn fourier_destroy_double(p: *mut FourierFftDouble);
        fn fourier_transform_in_place_double(p: *const FourierFftDouble, x: *mut Complex<f64>, t: c_int);
        fn fourier_transform_double(p: *const FourierFftDouble, i: *const Complex<f64>, o: *mut Complex<f64>, t: c_int);
        // Part 2: device-resident / host-streamed batches, status
        fn fourier_hip_create_float(size: usize, device: c_int) -> *mut FourierFftFloat;
        fn fourier_hip_create_double(size: usize, device: c_int) -> *mut FourierFftDouble;
        fn fourier_hip_transform_batch_float(p: *const FourierFftFloat, d_in: *const c_void, d_out: *mut c_void,
                                             batch: usize, t: c_int, stream: *mut c_void) -> c_int;
        fn fourier_hip_transform_batch_double(p: *const FourierFftDouble, d_in: *const c_void, d_out: *mut c_void,
                                              batch: usize, t: c_int, stream: *mut c_void) -> c_int;
        fn fourier_hip_transform_batch_host_float(p: *const FourierFftFloat, i: *const Complex<f32>, o: *mut Complex<f32>,
                                                  batch: usize, t: c_int) -> c_int;
        fn fourier_hip_transform_batch_host_double(p: *const FourierFftDouble, i: *const Complex<f64>, o: *mut Complex<f64>,
                                                   batch: usize, t: c_int) -> c_int;
        fn fourier_hip_reserve_float(p: *const FourierFftFloat, batch: usize, in_place: c_int) -> c_int;
        fn fourier_hip_reserve_double(p: *const FourierFftDouble, batch: usize, in_place: c_int) -> c_int;
        fn fourier_hip_synchronize_float(p: *const FourierFftFloat, stream: *mut c_void) -> c_int;
        fn fourier_hip_synchronize_double(p: *const FourierFftDouble, stream: *mut c_void) -> c_int;
        fn fourier_hip_last_status_float(p: *const FourierFftFloat) -> c_int;
        fn fourier_hip_last_status_double(p: *const FourierFftDouble) -> c_int;
        fn fourier_hip_status_string(status: c_int) -> *const c_char;
        fn fourier_hip_set_default_option(key: *const c_char, value: i64) -> c_int;
    }

    /// Inverse of `convert_transform` (fourier-ffi/src/lib.rs:3-12).
    pub(crate) fn code(t: Transform) -> c_int {
        match t {
            Transform::Fft => 0,
            Transform::Ifft => 1,
            Transform::UnscaledIfft => 2,
            Transform::SqrtScaledFft => 3,
            Transform::SqrtScaledIfft => 4,
        }
    }

    /// Error of a batched call: the engine's status code and its text.
    #[derive(Debug, Clone, PartialEq, Eq)]
    pub struct HipError {
        pub status: i32,
        pub message: String,
    }
    fn check(status: c_int) -> Result<(), HipError> {
        if status == 0 {
            return Ok(());
        }
        let message = unsafe { std::ffi::CStr::from_ptr(fourier_hip_status_string(status)) }.to_string_lossy().into_owned();
        Err(HipError { status, message })
    }

    /// Library-wide policy for plans created afterwards (include/fourier.h, `fourier_hip_set_default_option`): what `create_fft_*`
    /// does for a length whose prime factors stop at 13 and that has no ahead-of-time route.
    #[derive(Debug, Clone, Copy, PartialEq, Eq)]
    pub enum SpecialiseAtCreate {
        /// run-time kernels are never picked up by `create`
        Never = 0,
        /// (the library's default) load them where the on-disk code-object cache holds them; nothing is compiled implicitly
        FromCache = 1,
        /// ... and compile what the cache lacks, inside `create` (about a second per new length and machine)
        Compile = 2,
    }
    /// One call at start-up gives every later `create_fft_f32/f64` the specialised kernels (the environment variable
    /// `FOURIER_HIP_SPECIALISE=0|1|2` does the same for a program that cannot be changed).
    pub fn set_specialise_at_create(policy: SpecialiseAtCreate) -> Result<(), HipError> {
        check(unsafe { fourier_hip_set_default_option(b"specialise_at_create\0".as_ptr() as *const c_char, policy as i64) })
    }

    macro_rules! hip_plan {
        ($name:ident, $real:ty, $handle:ty, $create:ident, $create_dev:ident, $destroy:ident, $in_place:ident, $oop:ident,
         $batch:ident, $batch_host:ident, $reserve:ident, $status:ident, $sync:ident) => {
            /// One plan on one device.  `Send`, not `Sync`.
            pub struct $name {
                handle: *mut $handle,
                size: usize,
            }
            // the engine is free-threaded across handles; one thread at a time per handle (like RefCell scratch)
            unsafe impl Send for $name {}

            impl $name {
                /// `None` where the reference's constructor would panic (the FFI returns NULL, lib.rs:18-19),
                /// and for size 0 (which hangs the reference, autosort/mod.rs:112-115).
                pub fn new(size: usize) -> Option<Self> {
                    let handle = unsafe { $create(size) };
                    if handle.is_null() { None } else { Some(Self { handle, size }) }
                }
                /// The same plan on a given HIP device.
                pub fn on_device(size: usize, device: i32) -> Option<Self> {
                    let handle = unsafe { $create_dev(size, device as c_int) };
                    if handle.is_null() { None } else { Some(Self { handle, size }) }
                }
                /// `batch` contiguous transforms in DEVICE memory, enqueued on `stream` (a `hipStream_t`, null = default).
                /// # Safety
                /// `d_in` / `d_out` must be device pointers to `batch * size` elements on the plan's device.
                pub unsafe fn transform_batch_device(&self, d_in: *const c_void, d_out: *mut c_void, batch: usize,
                                                     transform: Transform, stream: *mut c_void) -> Result<(), HipError> {
                    check($batch(self.handle, d_in, d_out, batch, code(transform), stream))
                }
                /// Many transforms held in host memory, streamed through the device (copies and kernels overlap).
                pub fn transform_batch(&self, input: &[Complex<$real>], output: &mut [Complex<$real>],
                                       transform: Transform) -> Result<(), HipError> {
                    assert_eq!(input.len(), output.len());
                    assert_eq!(input.len() % self.size, 0);
                    check(unsafe { $batch_host(self.handle, input.as_ptr(), output.as_mut_ptr(), input.len() / self.size, code(transform)) })
                }
                /// Pre-size the plan's device buffers so that later device batches of up to `batch` never allocate.
                pub fn reserve(&self, batch: usize, in_place: bool) -> Result<(), HipError> {
                    check(unsafe { $reserve(self.handle, batch, in_place as c_int) })
                }
                /// Wait for everything queued on `stream` (null = the NULL stream) of the plan's device: the blocking half of
                /// `transform_batch_device` for a caller that has no HIP runtime binding of its own.
                /// # Safety
                /// `stream` must be null or a live `hipStream_t` of the plan's device.
                pub unsafe fn synchronize(&self, stream: *mut c_void) -> Result<(), HipError> {
                    check($sync(self.handle, stream))
                }
                fn last_status(&self) -> c_int {
                    unsafe { $status(self.handle) }
                }
            }

            impl Drop for $name {
                fn drop(&mut self) {
                    unsafe { $destroy(self.handle) }
                }
            }

            impl Fft for $name {
                type Real = $real;

                fn size(&self) -> usize {
                    self.size
                }

                fn transform_in_place(&self, input: &mut [Complex<$real>], transform: Transform) {
                    assert_eq!(input.len(), self.size); // autosort/mod.rs:332-333, bluesteins.rs:226
                    unsafe { $in_place(self.handle, input.as_mut_ptr(), code(transform)) }
                    // the legacy calls are `void`; surface a device failure the way the pure-Rust plans surface
                    // theirs (a panic), instead of returning stale data
                    let status = self.last_status();
                    assert!(status == 0, "libfourier: status {}", status);
                }

                fn transform(&self, input: &[Complex<$real>], output: &mut [Complex<$real>], transform: Transform) {
                    assert_eq!(input.len(), self.size); // fft.rs:57-58
                    assert_eq!(output.len(), self.size);
                    unsafe { $oop(self.handle, input.as_ptr(), output.as_mut_ptr(), code(transform)) }
                    let status = self.last_status();
                    assert!(status == 0, "libfourier: status {}", status);
                }
            }
        };
    }

    hip_plan!(HipFft32, f32, FourierFftFloat, fourier_create_float, fourier_hip_create_float, fourier_destroy_float,
              fourier_transform_in_place_float, fourier_transform_float, fourier_hip_transform_batch_float,
              fourier_hip_transform_batch_host_float, fourier_hip_reserve_float, fourier_hip_last_status_float,
              fourier_hip_synchronize_float);
    hip_plan!(HipFft64, f64, FourierFftDouble, fourier_create_double, fourier_hip_create_double, fourier_destroy_double,
              fourier_transform_in_place_double, fourier_transform_double, fourier_hip_transform_batch_double,
              fourier_hip_transform_batch_host_double, fourier_hip_reserve_double, fourier_hip_last_status_double,
              fourier_hip_synchronize_double);
}

#[cfg(feature = "hip")]
pub use hip::{set_specialise_at_create, HipError, HipFft32, HipFft64, SpecialiseAtCreate};

/// Create a complex-valued FFT over `f32` with the specified size (signature of `fourier/src/lib.rs:31`).
#[cfg(feature = "hip")]
pub fn create_fft_f32(size: usize) -> Box<dyn Fft<Real = f32> + Send> {
    Box::new(HipFft32::new(size).expect("cannot create FFT plan"))
}

/// Create a complex-valued FFT over `f64` with the specified size (signature of `fourier/src/lib.rs:49`).
#[cfg(feature = "hip")]
pub fn create_fft_f64(size: usize) -> Box<dyn Fft<Real = f64> + Send> {
    Box::new(HipFft64::new(size).expect("cannot create FFT plan"))
}

/// Without the `hip` feature the crate is a pass-through to the pure-Rust plans.
#[cfg(not(feature = "hip"))]
pub use fourier::{create_fft_f32, create_fft_f64};
