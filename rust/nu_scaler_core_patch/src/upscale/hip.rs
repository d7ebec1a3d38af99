//! `HipUpscaler`: the unchanged `trait Upscaler` of `upscale/mod.rs:67-88` over libnuscaler_hip.so (MI355X / gfx950).
//! Wiring: `UpscalerFactory::create_upscaler` (`upscale/mod.rs:95-117`) returns `Box::new(HipUpscaler::new(..))`
//! where it returns `WgpuUpscaler` today; `PyWgpuUpscaler` (`lib.rs:39-160`) holds `inner: HipUpscaler`.
use super::{UpscaleAlgorithm, Upscaler, UpscalingQuality};
use anyhow::{anyhow, Result};
use nu_scaler_hip_sys as sys;
use std::{any::Any, ffi::CStr};

pub struct HipUpscaler {
    h: *mut sys::nus_upscaler,
    quality: UpscalingQuality,
}

// One C handle serialises concurrent calls with its own mutex: rayon calls `upscale(&self)` from several
// threads (`upscale/mod.rs:619-624`), and the pyclass that owns it is `Send` (`lib.rs:39`).
unsafe impl Send for HipUpscaler {}
unsafe impl Sync for HipUpscaler {}

fn quality_code(q: UpscalingQuality) -> i32 {
    match q {
        UpscalingQuality::UltraPerformance => sys::NUS_QUALITY_ULTRA_PERFORMANCE,
        UpscalingQuality::Ultra => sys::NUS_QUALITY_ULTRA,
        UpscalingQuality::Quality => sys::NUS_QUALITY_QUALITY,
        UpscalingQuality::Balanced => sys::NUS_QUALITY_BALANCED,
        UpscalingQuality::Performance => sys::NUS_QUALITY_PERFORMANCE,
        UpscalingQuality::Native => sys::NUS_QUALITY_NATIVE,
    }
}

impl HipUpscaler {
    pub fn new(quality: UpscalingQuality, algorithm: UpscaleAlgorithm) -> Self {
        let alg = match algorithm {
            UpscaleAlgorithm::Nearest => sys::NUS_ALG_NEAREST,
            UpscaleAlgorithm::Bilinear => sys::NUS_ALG_BILINEAR,
            // a `Lanczos3` variant added to the enum maps to sys::NUS_ALG_LANCZOS3
        };
        let h = unsafe { sys::nus_upscaler_create(alg, quality_code(quality)) };
        assert!(!h.is_null(), "nus_upscaler_create: {}", last_thread_error());
        Self { h, quality }
    }

    /// BGRA capture frames (`lib.rs:251-270` swizzles them on the CPU): swizzled inside the kernels' loads instead.
    pub fn set_bgra_input(&mut self, bgra: bool) -> Result<()> {
        let f = if bgra { sys::NUS_FORMAT_BGRA8 } else { sys::NUS_FORMAT_RGBA8 };
        self.check(unsafe { sys::nus_upscaler_set_input_format(self.h, f) })
    }

    /// Device-resident results to host bytes: what `WgpuUpscaler::upscale` does at its end (map the staging buffer, wait, `to_vec`:
    /// `upscale/mod.rs:1041-1057`) for callers of the `*_device` entry points.  `d_src` is a device pointer, `stream` the `hipStream_t`
    /// the frame was computed on (null: the device's null stream); the copy is ordered after that work.  The `Vec` is pageable memory
    /// and is never handed to the HIP runtime: the library stages through pinned chunks of its own (`nus_download`).
    pub fn download(d_src: *const std::ffi::c_void, bytes: usize, stream: *mut std::ffi::c_void) -> Result<Vec<u8>> {
        let mut out: Vec<u8> = Vec::with_capacity(bytes);
        let rc = unsafe { sys::nus_download(out.as_mut_ptr() as *mut std::ffi::c_void, d_src, bytes, stream) };
        if rc != sys::NUS_OK {
            return Err(anyhow!(last_thread_error()));
        }
        unsafe { out.set_len(bytes) };
        Ok(out)
    }

    /// Host bytes to a device buffer (`queue.write_buffer`, `upscale/mod.rs:968-1008`); `src` may be re-used on return.
    pub fn upload(d_dst: *mut std::ffi::c_void, src: &[u8], stream: *mut std::ffi::c_void) -> Result<()> {
        let rc = unsafe { sys::nus_upload(d_dst, src.as_ptr() as *const std::ffi::c_void, src.len(), stream) };
        if rc != sys::NUS_OK {
            return Err(anyhow!(last_thread_error()));
        }
        Ok(())
    }

    /// `WgpuUpscaler::upscale_batch` (`upscale/mod.rs:609-640`, an inherent method there too, called by
    /// `PyWgpuUpscaler::upscale_batch`, `lib.rs:140-154`).
    pub fn upscale_batch(&self, frames: &[&[u8]]) -> Result<Vec<Vec<u8>>> {
        // upscale/mod.rs:609-640 fans out over rayon; here the library pipelines H2D / kernel / D2H over its slot streams
        let n = unsafe { sys::nus_upscaler_output_size(self.h) };
        let mut outs: Vec<Vec<u8>> = frames.iter().map(|_| Vec::with_capacity(n)).collect();
        let ins: Vec<*const u8> = frames.iter().map(|f| f.as_ptr()).collect();
        let lens: Vec<usize> = frames.iter().map(|f| f.len()).collect();
        let out_ptrs: Vec<*mut u8> = outs.iter_mut().map(|o| o.as_mut_ptr()).collect();
        let rc = unsafe {
            sys::nus_upscaler_upscale_batch(self.h, ins.as_ptr(), lens.as_ptr(), frames.len(), out_ptrs.as_ptr(), n)
        };
        self.check(rc)?;
        for o in outs.iter_mut() {
            unsafe { o.set_len(n) };
        }
        Ok(outs)
    }

    fn err(&self) -> anyhow::Error {
        anyhow!(unsafe { CStr::from_ptr(sys::nus_upscaler_last_error(self.h)) }
            .to_string_lossy()
            .into_owned())
    }

    fn check(&self, rc: i32) -> Result<()> {
        if rc == sys::NUS_OK {
            Ok(())
        } else {
            Err(self.err())
        }
    }
}

fn last_thread_error() -> String {
    unsafe { CStr::from_ptr(sys::nus_last_error()) }.to_string_lossy().into_owned()
}

impl Drop for HipUpscaler {
    fn drop(&mut self) {
        unsafe { sys::nus_upscaler_destroy(self.h) }
    }
}

impl Upscaler for HipUpscaler {
    fn initialize(&mut self, input_width: u32, input_height: u32, output_width: u32, output_height: u32) -> Result<()> {
        self.check(unsafe { sys::nus_upscaler_initialize(self.h, input_width, input_height, output_width, output_height) })
    }

    fn upscale(&self, input: &[u8]) -> Result<Vec<u8>> {
        // "Upscaler not initialized. Call initialize() first." and the size-mismatch text of upscale/mod.rs:937-966
        // come back verbatim from the library.
        let n = unsafe { sys::nus_upscaler_output_size(self.h) };
        let mut out = Vec::<u8>::with_capacity(n);
        let rc = unsafe { sys::nus_upscaler_upscale(self.h, input.as_ptr(), input.len(), out.as_mut_ptr(), n) };
        self.check(rc)?;
        unsafe { out.set_len(n) };
        Ok(out)
    }

    fn name(&self) -> &'static str {
        // "WgpuNearestUpscaler" / "WgpuBilinearUpscaler" as upscale/mod.rs:1060-1066; the strings are static in the library
        unsafe { CStr::from_ptr(sys::nus_upscaler_name(self.h)) }.to_str().unwrap_or("HipUpscaler")
    }

    fn quality(&self) -> UpscalingQuality {
        self.quality
    }

    fn set_quality(&mut self, quality: UpscalingQuality) -> Result<()> {
        self.quality = quality;
        self.check(unsafe { sys::nus_upscaler_set_quality(self.h, quality_code(quality)) })
    }

    fn as_any(&self) -> &dyn Any {
        self
    }

    fn as_any_mut(&mut self) -> &mut dyn Any {
        self
    }
}
