"""Consumer side of /anchor_correction without ROS.

Mirrors how the reference's pose relay uses the matcher's anchors
(simulation/isaac/scripts/common/tf_wall_clock_relay_v55.py:193-199 thresholds, :235-256 callback,
:533-591 regime switch and blend).  Only x, y and covariance[0] of the message are ever read.  Scalar
logic at 20 Hz: host Python, no device work; it exists so a teach -> repeat -> fusion replay can be tested
end to end without ROS.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

ANCHOR_STALE_S = 3.0       # older anchors are ignored
ANCHOR_STRONG_STD = 0.1    # std at or below: "strong"
ANCHOR_OK_STD = 0.2        # std at or below: usable with the weaker blend
ANCHOR_HYSTERESIS_N = 2    # consecutive strong anchors needed for the strong regime

WEIGHTS = {"strong": (0.40, 0.55, 0.05), "ok": (0.20, 0.75, 0.05)}   # anchor, SLAM, encoder


@dataclass
class FusedPose:
    x: float
    y: float
    regime: str
    alpha: float | None = None         # SLAM weight in the no_anchor regime
    anchor_staleness: float = -1.0
    anchor_std: float = -1.0


class AnchorFusion:
    def __init__(self):
        self.anchor_last = None        # (ts, x, y, std)
        self.anchor_strong_streak = 0

    def on_anchor(self, ts: float, x: float, y: float, cov0: float):
        """the relay's _anchor_cb: std = sqrt(max(cov[0], 1e-8)); streak up on strong, down (floored) otherwise"""
        std = math.sqrt(max(cov0, 1e-8))
        self.anchor_last = (ts, x, y, std)
        if std <= ANCHOR_STRONG_STD:
            self.anchor_strong_streak += 1
        else:
            self.anchor_strong_streak = max(0, self.anchor_strong_streak - 1)
        return std

    def on_outcome(self, outcome):
        """feed a matcher TickOutcome (published ones carry pose and covariance)"""
        if outcome is not None and outcome.published:
            self.on_anchor(outcome.ts, outcome.anchor_pose[0], outcome.anchor_pose[1], outcome.covariance[0])

    def regime(self, now: float):
        if self.anchor_last is None:
            return "no_anchor"
        ts, _, _, std = self.anchor_last
        if now - ts <= ANCHOR_STALE_S and std <= ANCHOR_OK_STD:
            if std <= ANCHOR_STRONG_STD and self.anchor_strong_streak >= ANCHOR_HYSTERESIS_N:
                return "strong"
            return "ok"
        return "no_anchor"

    def blend(self, now: float, slam_xy, enc_xy) -> FusedPose:
        """nav position from SLAM, encoder and (when fresh) the anchor"""
        reg = self.regime(now)
        stale = (now - self.anchor_last[0]) if self.anchor_last else -1.0
        std = self.anchor_last[3] if self.anchor_last else -1.0
        if reg in WEIGHTS:
            wa, ws, we = WEIGHTS[reg]
            ax, ay = self.anchor_last[1], self.anchor_last[2]
            return FusedPose(wa * ax + ws * slam_xy[0] + we * enc_xy[0], wa * ay + ws * slam_xy[1] + we * enc_xy[1], reg,
                             None, stale, std)
        d = math.hypot(slam_xy[0] - enc_xy[0], slam_xy[1] - enc_xy[1])
        age = stale if self.anchor_last else 999
        if age > 10.0:
            alpha = 0.95 if d < 2.0 else 0.70 if d < 5.0 else 0.40 if d < 10.0 else 0.10
        else:
            alpha = 0.95
        return FusedPose(alpha * slam_xy[0] + (1 - alpha) * enc_xy[0], alpha * slam_xy[1] + (1 - alpha) * enc_xy[1],
                         "no_anchor", alpha, stale, std)
