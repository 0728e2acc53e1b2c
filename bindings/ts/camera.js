'use strict';
/*
 * camera.js (+ camera.d.ts) -- the reference's `Camera` (src/camera/camera.ts:100-205) without the browser: `canvas` is any object with
 * `width` and `height` (an HTMLCanvasElement has them), `device` a HipDevice.  The camera owns the 272-byte uniform buffer the forward pass
 * reads (view, view_inv, proj, proj_inv, viewport, focal) and rewrites it, stream-ordered, on every update_buffer() -- queue.writeBuffer in
 * the reference (camera.ts:194).  Mouse / keyboard control (camera-control.ts) is UI and stays out (SURVEY 2.1 row 13).
 */
const { mat4Inverse, projectionMatrix } = require('./camera-math.js');

const UNIFORM_BYTES = 4 * 64 + 2 * 8;

/** create_camera_uniform_buffer (camera.ts:76-82). */
function create_camera_uniform_buffer(device) { return device.createBuffer({ label: 'camera uniform', size: UNIFORM_BYTES }); }

/** The 68 floats of CameraUniforms for a pose (column-major rotation, world position) and a projection given by its two fields of view. */
function uniformBlock(rotation, position, fovX, fovY, width, height, focal) {
  const out = new Float32Array(68);
  // get_view_matrix = mat4.translate(r, -t) (camera.ts:23-26): the rotation's columns, and as fourth column r * (-t, 1)
  out.set(rotation.subarray(0, 12), 0);
  const tx = -position[0], ty = -position[1], tz = -position[2];
  for (let r = 0; r < 4; r++) out[12 + r] = rotation[r] * tx + rotation[4 + r] * ty + rotation[8 + r] * tz + rotation[12 + r];
  out.set(projectionMatrix(0.01, 100, fovX, fovY), 32);
  out.set(mat4Inverse(out.subarray(0, 16)), 16);
  out.set(mat4Inverse(out.subarray(32, 48)), 48);
  out[64] = width; out[65] = height; out[66] = focal; out[67] = focal;
  return out;
}

function normalize3(v) { const n = Math.hypot(v[0], v[1], v[2]); return n > 1e-5 ? Float32Array.from([v[0] / n, v[1] / n, v[2] / n]) : new Float32Array(3); }

class Camera {
  constructor(canvas, device) {
    this.canvas = canvas; this.device = device;
    this.uniform_buffer = create_camera_uniform_buffer(device);
    this.position = new Float32Array(3); this.rotation = new Float32Array(16);
    this.focal = new Float32Array(2); this.viewport = new Float32Array(2);
    this.look = Float32Array.from([0, 0, 1]); this.up = Float32Array.from([0, 1, 0]); this.right = Float32Array.from([1, 0, 0]);
    this.uniforms = new Float32Array(68);   // host copy of what the buffer holds
    this.reset();
  }
  reset() {   // camera.ts:113-119: position (0, 0, 5), identity rotation, fovY 45 degrees
    this.position.set([0, 0, 5]);
    this.rotation.fill(0); this.rotation[0] = this.rotation[5] = this.rotation[10] = this.rotation[15] = 1;
    this.fovY = 45 / 180 * Math.PI; this.fovX = this.fovY;
    this.on_update_canvas();
  }
  on_update_canvas() {   // camera.ts:121-130
    const focal = 0.5 * this.canvas.height / Math.tan(this.fovY * 0.5);
    this.focal[0] = focal; this.focal[1] = focal;
    this.fovX = 2 * Math.atan(this.canvas.width / (2 * focal));
    this.viewport[0] = this.canvas.width; this.viewport[1] = this.canvas.height;
    this.update_buffer();
  }
  update_buffer() {   // camera.ts:165-195; the focal and viewport fields are the Float32Array values the reference copies
    this.uniforms = uniformBlock(this.rotation, this.position, this.fovX, this.fovY, this.viewport[0], this.viewport[1], this.focal[0]);
    const inv = this.uniforms.subarray(16, 32);   // look / right / up = the inverse view's upper 3x3 applied to the canonical axes
    this.look = normalize3([inv[8], inv[9], inv[10]]); this.right = normalize3([inv[0], inv[1], inv[2]]); this.up = normalize3([inv[4], inv[5], inv[6]]);
    this.device.queue.writeBuffer(this.uniform_buffer, 0, this.uniforms);
  }
  set_preset(preset) {   // camera.ts:196-205: CameraData (loaders.js)
    if (preset.position) this.position.set(Array.prototype.slice.call(preset.position, 0, 3));
    if (preset.rotation) this.rotation.set(preset.rotation);
    if (preset.fx && preset.fy && preset.height) this.fovY = 2 * Math.atan(preset.height / (2 * preset.fy));
    this.on_update_canvas();
  }
  destroy() { this.uniform_buffer.destroy(); }
}

/** load_camera_presets (camera.ts:63-89): [{ position, rotation }] from the camera JSON; mat3.create(...rotation.flat()) + mat4.fromMat3 read the
 *  nine numbers in file order as COLUMNS (the transpose of what loaders.loadCameraJson builds -- both are the reference's). */
function load_camera_presets(file) {
  const text = Buffer.isBuffer(file) ? file.toString('utf8') : (typeof file === 'string' ? file : Buffer.from(file).toString('utf8'));
  return JSON.parse(text).map((j) => {
    const v = [].concat.apply([], j.rotation), rotation = new Float32Array(16);
    for (let c = 0; c < 3; c++) for (let r = 0; r < 3; r++) rotation[c * 4 + r] = v[c * 3 + r];
    rotation[15] = 1;
    return { position: Float32Array.from(j.position.slice(0, 3)), rotation };
  });
}

module.exports = { Camera, create_camera_uniform_buffer, load_camera_presets, uniformBlock };
