"""Scripted fly-through: what a user does with the reference's window, as a script.

Every frame is `state.update(); demo.update_gpu_state(); demo.compute_pass()` (src/event_loop.rs:100-119) after one
scripted input event:
  * mouse drag / scroll wheel -> CameraController::process_mouse / process_scroll -> Camera::orbit
    (src/camera.rs:47-61, :85-117; sensitivity 0.2 degrees per pixel, zoom 0.2 per line, src/state.rs:63);
  * the egui panel's widgets (src/gui.rs:198-277) with their rules and ranges: "Importance Rendering" on forces "Opacity" on
    (:228-236) and the opacity box is disabled while it is on (:211-222); "Cone Importance Check" and "Look Ahead Steps"
    (2..=25) are enabled only with importance rendering (:242-262); "Raymarching Step Size" 0.001..=0.1 logarithmic
    (:264-269); "Density Threshold" 0.005..=1.0 (:271-275).
"""
import numpy as np

LOOK_AHEAD_RANGE = (2, 25)          # src/gui.rs:259
STEP_RANGE = (0.001, 0.1)           # src/gui.rs:265
THRESHOLD_RANGE = (0.005, 1.0)      # src/gui.rs:272


class Gui:
    """The widgets of src/gui.rs:198-277 acting on a State (volym_amd.scene.State)."""

    def __init__(self, state):
        self.s = state.c

    def importance_coloring(self, on):
        self.s.use_importance_coloring = 1 if on else 0                  # :199-203

    def opacity(self, on):
        if self.s.use_importance_rendering:                              # add_enabled(!use_importance_rendering, ..) :211-222
            return
        self.s.use_opacity = 1 if on else 0

    def importance_rendering(self, on):
        self.s.use_importance_rendering = 1 if on else 0                 # :225-231
        if on:
            self.s.use_opacity = 1                                       # :233-235

    def cone_importance_check(self, on):
        if not self.s.use_importance_rendering:                          # add_enabled(use_importance_rendering, ..) :242-251
            return
        self.s.use_cone_importance_check = 1 if on else 0

    def gaussian_smoothing(self, on):
        self.s.use_gaussian_smoothing = 1 if on else 0                   # :253-257

    def look_ahead_steps(self, n):
        if not self.s.use_importance_rendering:                          # :257-262
            return
        self.s.importance_check_ahead_steps = int(min(max(int(n), LOOK_AHEAD_RANGE[0]), LOOK_AHEAD_RANGE[1]))

    def step_size(self, v):
        self.s.raymarching_step_size = float(min(max(float(v), STEP_RANGE[0]), STEP_RANGE[1]))            # :264-269

    def density_threshold(self, v):
        self.s.density_threshold = float(min(max(float(v), THRESHOLD_RANGE[0]), THRESHOLD_RANGE[1]))     # :271-275


def script(n_frames, seed=20250310):
    """Deterministic list of per-frame events: ("mouse", dx, dy) | ("scroll", lines) | (widget name, value).
    A slow orbit with zooms, and every widget exercised over its range (the extremes of the sliders included)."""
    rng = np.random.default_rng(seed)
    events = []
    widgets = [("importance_rendering", 1), ("look_ahead_steps", 2), ("cone_importance_check", 1), ("look_ahead_steps", 25),
               ("opacity", 0), ("cone_importance_check", 0), ("importance_rendering", 0), ("opacity", 0), ("step_size", 0.1),
               ("gaussian_smoothing", 1), ("step_size", 0.02), ("density_threshold", 0.005), ("importance_coloring", 1),
               ("density_threshold", 0.3), ("importance_coloring", 0), ("opacity", 1), ("gaussian_smoothing", 0),
               ("step_size", 0.004), ("density_threshold", 1.0), ("density_threshold", 0.15), ("importance_rendering", 1),
               ("look_ahead_steps", 15), ("step_size", 0.01)]
    w = 0
    for i in range(n_frames):
        if i % 3 == 2 and w < len(widgets):
            events.append(widgets[w])
            w += 1
        elif i % 11 == 10:
            events.append(("scroll", float(rng.choice([-3.0, -1.0, 1.0, 2.0]))))
        else:
            events.append(("mouse", float(rng.uniform(-60.0, -5.0)), float(rng.uniform(-12.0, 12.0))))
    return events


def apply(state, event):
    """One scripted input event -> State (the window's input handlers and the egui panel)."""
    kind = event[0]
    if kind == "mouse":
        state.process_mouse(event[1], event[2])
    elif kind == "scroll":
        state.process_scroll(event[1])
    else:
        getattr(Gui(state), kind)(event[1])
