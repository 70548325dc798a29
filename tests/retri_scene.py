"""A small moving stereo rig observing static points: the input stream of the retriangulation tests (CPU oracle and GPU)."""
import numpy as np

FX, FY, CX, CY, W, H = 458.0, 457.0, 367.0, 248.0, 752, 480


def rot_z(a):
    return np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])


class Scene:
    def __init__(self, T=9, n_pts=40, seed=3, noise=0.0, p_see=(0.9, 0.8)):
        rng = np.random.default_rng(seed)
        self.T, self.K = T, 2
        # IMU poses: R_GtoI, p_IinG — forward motion with a slow yaw; cameras look along the IMU z axis
        self.R_GtoI = [rot_z(0.03 * t).T for t in range(T)]
        self.p_IinG = [np.array([0.25 * t, 0.05 * np.sin(t), 0.02 * t]) for t in range(T)]
        self.R_ItoC = [np.eye(3), rot_z(0.01)]
        self.p_IinC = [np.zeros(3), np.array([-0.11, 0.0, 0.0])]
        self.pts = np.column_stack([rng.uniform(-2.5, 2.5, n_pts), rng.uniform(-1.5, 1.5, n_pts), rng.uniform(3.0, 8.0, n_pts)]) + np.array([1.0, 0, 0])
        self.obs = []  # per frame: dict cam -> list of (featid, (u, v) f32, (xn, yn) f32)
        for t in range(T):
            fr = {0: [], 1: []}
            for k in range(2):
                R_GtoC = self.R_ItoC[k] @ self.R_GtoI[t]
                for i, p in enumerate(self.pts):
                    if rng.uniform() > p_see[k]:
                        continue
                    pc = R_GtoC @ (p - self.p_IinG[t]) + self.p_IinC[k]
                    if pc[2] < 0.2:
                        continue
                    xn, yn = pc[0] / pc[2] + noise * rng.normal(), pc[1] / pc[2] + noise * rng.normal()
                    u, v = FX * xn + CX, FY * yn + CY
                    fr[k].append((100 + i, (np.float32(u), np.float32(v)), (np.float32(xn), np.float32(yn))))
            self.obs.append(fr)

    def cams(self):
        return [(k, self.R_ItoC[k], self.p_IinC[k]) for k in range(2)]

    def pose_table(self):
        """R_GtoC [K T 9], p_CinG [K T 3] for ovgpu_set_camera_poses (index k T + t)."""
        R = np.zeros((self.K * self.T, 9))
        p = np.zeros((self.K * self.T, 3))
        for k in range(self.K):
            for t in range(self.T):
                R_GtoC = self.R_ItoC[k] @ self.R_GtoI[t]
                R[k * self.T + t] = R_GtoC.reshape(-1)
                p[k * self.T + t] = self.p_IinG[t] - R_GtoC.T @ self.p_IinC[k]
        return np.ascontiguousarray(R), np.ascontiguousarray(p)

    def flat(self, t):
        """The frame's observations as the flat arrays of ovgpu_retriangulate (camera 0 first, then camera 1)."""
        ids, cams, uv, uvn = [], [], [], []
        for k in range(2):
            for fid, pd, pn in self.obs[t][k]:
                ids.append(fid), cams.append(k), uv.append(pd), uvn.append(pn)
        return np.array(ids, np.int64), np.array(cams, np.int32), np.array(uv, np.float32).reshape(-1, 2), np.array(uvn, np.float32).reshape(-1, 2)
