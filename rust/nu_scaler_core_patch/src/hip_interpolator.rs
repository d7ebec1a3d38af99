//! The Python-facing interpolator of `wgpu_interpolator.rs:169-498` over libnuscaler_hip.so: same class name,
//! constructor argument, method names, keyword (`time_t`), return types and error texts, so
//! `nu_scaler_py/nu_scaler/main.py:999-1005` needs no change.  The reference always warps with a zero flow field
//! (`wgpu_interpolator.rs:275-295`); `flow = NULL` asks the library for exactly that.
use nu_scaler_hip_sys as sys;
use pyo3::exceptions::{PyRuntimeError, PyValueError};
use pyo3::prelude::*;
use pyo3::types::PyBytes;
use std::ffi::CStr;

#[pyclass(name = "WgpuFrameInterpolator")]
pub struct HipFrameInterpolator {
    h: *mut sys::nus_interp,
}

// one C handle serialises concurrent calls with its own mutex
unsafe impl Send for HipFrameInterpolator {}
unsafe impl Sync for HipFrameInterpolator {}

fn preset_code(s: Option<&str>) -> i32 {
    // WorkgroupSizePreset::from_string (wgpu_interpolator.rs:98-127); anything else: the reference's default, Wide32x8
    match s.map(|v| v.to_ascii_lowercase()) {
        Some(ref v) if v == "square8x8" || v == "8x8" => sys::NUS_WG_SQUARE_8X8,
        Some(ref v) if v == "square16x16" || v == "16x16" => sys::NUS_WG_SQUARE_16X16,
        Some(ref v) if v == "tall8x32" || v == "8x32" => sys::NUS_WG_TALL_8X32,
        _ => sys::NUS_WG_WIDE_32X8,
    }
}

#[pymethods]
impl HipFrameInterpolator {
    #[new]
    #[pyo3(signature = (workgroup_preset_str=None))]
    fn new_py(workgroup_preset_str: Option<String>) -> PyResult<Self> {
        let h = unsafe { sys::nus_interp_create(preset_code(workgroup_preset_str.as_deref())) };
        if h.is_null() {
            let msg = unsafe { CStr::from_ptr(sys::nus_last_error()) }.to_string_lossy().into_owned();
            return Err(PyRuntimeError::new_err(format!("Failed to initialize WgpuFrameInterpolator internals: {}", msg)));
        }
        Ok(Self { h })
    }

    #[pyo3(signature = (frame_a_bytes, frame_b_bytes, width, height, *, time_t=0.5))]
    fn interpolate_py<'py>(
        &self,
        py: Python<'py>,
        frame_a_bytes: &'py PyBytes,
        frame_b_bytes: &'py PyBytes,
        width: u32,
        height: u32,
        time_t: f32,
    ) -> PyResult<Py<PyBytes>> {
        let expected_size = (width as usize) * (height as usize) * 4;
        let (a, b) = (frame_a_bytes.as_bytes(), frame_b_bytes.as_bytes());
        if a.len() != expected_size || b.len() != expected_size {
            // the text of wgpu_interpolator.rs:234-237
            return Err(PyValueError::new_err(format!(
                "Expected {} bytes per frame for {}x{}x4 RGBA, got frame_a: {} bytes, frame_b: {} bytes",
                expected_size, width, height, a.len(), b.len()
            )));
        }
        let h = self.h as usize; // raw pointers are not Send: carry the address across allow_threads
        let result = PyBytes::new_bound_with(py, expected_size, |out: &mut [u8]| {
            let (ap, al, bp, bl, op, ol) = (a.as_ptr() as usize, a.len(), b.as_ptr() as usize, b.len(), out.as_mut_ptr() as usize, out.len());
            // the reference holds the GIL through the whole call (lib.rs:105-112); nothing here needs it
            let rc = py.allow_threads(move || unsafe {
                sys::nus_interp_interpolate(
                    h as *mut sys::nus_interp,
                    ap as *const u8,
                    al,
                    bp as *const u8,
                    bl,
                    std::ptr::null(), // zero flow
                    width,
                    height,
                    time_t,
                    op as *mut u8,
                    ol,
                )
            });
            if rc != sys::NUS_OK {
                let msg = unsafe { CStr::from_ptr(sys::nus_interp_last_error(h as *const sys::nus_interp)) }
                    .to_string_lossy()
                    .into_owned();
                return Err(PyRuntimeError::new_err(msg));
            }
            Ok(())
        })?;
        Ok(result.into())
    }

    // -- the shape of `trait FrameInterpolator` (`interpolation/mod.rs:29-44`): initialize / interpolate / name / quality
    fn initialize(&self, width: u32, height: u32) -> PyResult<()> {
        let rc = unsafe { sys::nus_interp_initialize(self.h, width, height) };
        if rc != sys::NUS_OK {
            let msg = unsafe { CStr::from_ptr(sys::nus_interp_last_error(self.h)) }.to_string_lossy().into_owned();
            return Err(PyRuntimeError::new_err(msg));
        }
        Ok(())
    }

    fn interpolate<'py>(&self, py: Python<'py>, frame1: &'py PyBytes, frame2: &'py PyBytes, t: f32) -> PyResult<Py<PyBytes>> {
        let (a, b) = (frame1.as_bytes(), frame2.as_bytes());
        let h = self.h;
        let result = PyBytes::new_bound_with(py, a.len(), |out: &mut [u8]| {
            let rc = unsafe { sys::nus_interp_interpolate_frames(h, a.as_ptr(), a.len(), b.as_ptr(), b.len(), t, out.as_mut_ptr(), out.len()) };
            if rc != sys::NUS_OK {
                let msg = unsafe { CStr::from_ptr(sys::nus_interp_last_error(h)) }.to_string_lossy().into_owned();
                return Err(PyRuntimeError::new_err(msg)); // "Interpolator not initialized" before initialize()
            }
            Ok(())
        })?;
        Ok(result.into())
    }

    #[getter]
    fn name(&self) -> String {
        unsafe { CStr::from_ptr(sys::nus_interp_name(self.h)) }.to_string_lossy().into_owned()
    }

    fn set_quality(&self, quality: i32) -> PyResult<()> {
        if unsafe { sys::nus_interp_set_quality(self.h, quality) } != sys::NUS_OK {
            return Err(PyValueError::new_err("unknown interpolation quality"));
        }
        Ok(())
    }

    #[getter]
    fn quality(&self) -> i32 {
        unsafe { sys::nus_interp_quality(self.h) }
    }

    /// `get_last_gpu_duration_ms` (`wgpu_interpolator.rs:494-497`): hipEvent time of the last warp + blend launch.
    fn get_last_gpu_duration_ms(&self) -> Option<f64> {
        let mut ms = 0.0f64;
        let rc = unsafe { sys::nus_interp_last_gpu_ms(self.h, &mut ms) };
        if rc == sys::NUS_OK {
            Some(ms)
        } else {
            None
        }
    }
}

impl Drop for HipFrameInterpolator {
    fn drop(&mut self) {
        unsafe { sys::nus_interp_destroy(self.h) }
    }
}
