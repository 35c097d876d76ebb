"""PRM with the edge checks of a whole roadmap in one device call.

Reference: numbotics/planning/sampling_based/{space.py:8-46, base.py:15-72, planners/prm.py:13-47,
graph.py:32-72,158-233}.  The reference's loop adds one sample per iteration, asks faiss for its k nearest
vertices and calls ``connector.connect`` once per neighbour (``prm.py:35-47``) -- ``max_iters * k_nearest``
sequential edge walks, each ``ceil(d/res)+1`` PyBullet queries.  The roadmap those iterations build depends only
on the sample sequence: vertex i is connected to its k nearest among the vertices that existed when it was added
(itself included, which ``connect`` rejects as a zero-length edge).  ``PRM.plan`` therefore draws the samples
exactly as ``SamplingPlannerBase.sample_state`` does, finds every neighbour list with an exact float32 L2 scan
(what ``faiss.IndexFlatL2`` computes; faiss itself is a third-party dependency, tie-breaking parity unpinned; on the
device when there is one: ``nbk_knn_prefix``),
and hands ALL candidate edges to ``DiscreteConnector.connect_batch`` -- one launch sequence on the device.

Only what PRM needs is here: the state space, the planner parameters, the roadmap (arrays + a SciPy Dijkstra
instead of networkx).  RRT grows a tree one steer at a time and stays sequential (``steer`` already checks a
whole edge per call); RRT* batches the neighbour connects of each iteration (``RRTStar`` below).
"""
from abc import ABC, abstractmethod
from dataclasses import dataclass
from random import choice
from typing import Callable, List, Optional

import numpy as np


class StateSpace(ABC):
    """space.py:8-46."""

    def __init__(self, lower_bounds, upper_bounds, sampler: Optional[Callable] = None):
        self._lower_bounds = lower_bounds
        self._upper_bounds = upper_bounds
        if self._lower_bounds.shape != self._upper_bounds.shape:
            raise ValueError("Lower and upper bounds must have the same shape")
        if self._lower_bounds.ndim != 1:
            raise ValueError("Lower and upper bounds must be 1D arrays")
        self._sampler = sampler

    def sample(self):
        if self._sampler is None:
            return np.random.uniform(self.lower_bounds, self.upper_bounds)
        return self._sampler()

    @abstractmethod
    def distance(self, state1, state2):
        raise NotImplementedError

    def distance_batch(self, states1, states2):
        """Row-wise ``distance``; subclasses may vectorise it as long as every row equals the scalar call bit for bit
        (the edge sampling step is ``resolution / distance``)."""
        return np.array([self.distance(a, b) for a, b in zip(states1, states2)], dtype=np.float64)

    @property
    def lower_bounds(self):
        return self._lower_bounds

    @property
    def upper_bounds(self):
        return self._upper_bounds

    @property
    def dimension(self):
        return self._lower_bounds.shape[0]

    @property
    def volume(self):
        return np.prod(self._upper_bounds - self._lower_bounds)


class EuclideanSpace(StateSpace):
    """L2 metric.  The scalar form goes through the same row reduction as the batched one, so both give identical bits."""

    def distance(self, state1, state2):
        return float(self.distance_batch(np.asarray(state1)[None], np.asarray(state2)[None])[0])

    def distance_batch(self, states1, states2):
        d = np.ascontiguousarray(np.asarray(states1, dtype=np.float64) - np.asarray(states2, dtype=np.float64))
        return np.sqrt((d * d).sum(axis=-1))


@dataclass(frozen=True)
class PlannerParams:
    """base.py:15-21."""
    max_iters: int
    goal_bias: float = 0.1
    rewire_factor: float = 1.1
    k_nearest: int = 50
    goal_tolerance: float = 1e-6


@dataclass(frozen=True)
class Node:
    """graph.py:10-19."""
    id: str
    state: np.ndarray
    cost: float = np.inf


def knn_prefix(points: np.ndarray, k: int, chunk: int = 1024, device=None):
    """For every i: the indices of the (at most k) nearest points among points[0..i] -- the neighbour lists an
    insert-then-query loop over an exact flat L2 index yields.  -> (N, k) int64, -1 padded.

    Distance = sum over the dimensions, in order, of (x - y)^2 in float32 (separate roundings); order = ascending
    (distance, index).  ``device``: None = the GPU kernel (``nbk_knn_prefix``) when a GPU is visible and k <= 64, else
    NumPy; True / False force one.  Both give identical lists."""
    x = np.ascontiguousarray(points, dtype=np.float32)
    n, dim = x.shape
    if device is None:
        device = False
        if k <= 64 and dim <= 64 and n > 0:
            try:
                import torch
                device = torch.cuda.is_available()
            except Exception:
                device = False
    if device:
        from numbotics_amd.engine import knn_prefix_device
        return knn_prefix_device(x, k)
    out = np.full((n, k), -1, dtype=np.int64)
    for a in range(0, n, chunk):
        b = min(n, a + chunk)
        d = np.zeros((b - a, b), dtype=np.float32)
        for c in range(dim):
            t = x[a:b, None, c] - x[None, :b, c]
            d += t * t
        d[np.arange(b - a)[:, None] + a < np.arange(b)[None, :]] = np.inf        # only points that already exist
        kk = min(k, b)
        # the kk smallest of every row without sorting the row: partition, then resolve ties at the k-th value
        # towards the smaller index, then order the selection by (distance, index)
        kth = np.partition(d, kk - 1, axis=1)[:, kk - 1]
        lt = d < kth[:, None]
        eq = d == kth[:, None]
        need = kk - lt.sum(axis=1)
        if (eq.sum(axis=1) == need).all():
            sel = lt | eq
        else:
            sel = lt | (eq & (np.cumsum(eq, axis=1, dtype=np.int32) <= need[:, None]))
        cols = np.nonzero(sel)[1].reshape(b - a, kk)                       # ascending index inside a row
        dsel = np.take_along_axis(d, cols, axis=1)
        order = np.argsort(dsel, axis=1, kind="stable")
        idx = np.take_along_axis(cols, order, axis=1)
        valid = np.take_along_axis(dsel, order, axis=1) < np.inf
        out[a:b, :kk] = np.where(valid, idx, -1)
    return out


class PRM:
    """prm.py:13-47 with the per-neighbour ``connect`` calls of all iterations batched."""

    def __init__(self, space: StateSpace, connector, params: PlannerParams):
        self._space = space
        self._connector = connector
        self._params = params
        self._start = None
        self._goals: List[np.ndarray] = []
        self.states = None           # (V, d) vertex states, v_0 = start
        self.edges = None            # (M, 2) int64 into the node list [vertices..., goals...]
        self.weights = None          # (M,)
        self.n_candidate_edges = 0

    # base.py:45-71
    def add_start(self, start):
        if not self._connector.is_valid(start):
            raise ValueError("Start state is invalid")
        if not np.all(start >= self._space.lower_bounds) or not np.all(start <= self._space.upper_bounds):
            raise ValueError("Start state is out of bounds")
        self._start = start

    def add_goal(self, goal):
        if not self._connector.is_valid(goal):
            raise ValueError("Goal state is invalid")
        if not np.all(goal >= self._space.lower_bounds) or not np.all(goal <= self._space.upper_bounds):
            raise ValueError("Goal state is out of bounds")
        self._goals.append(goal)

    def sample_state(self):
        if self._start is None:
            raise ValueError("Start state not set")
        if len(self._goals) == 0:
            raise ValueError("Goal states not set")
        return choice(self._goals) if np.random.rand() < self._params.goal_bias else self._space.sample()

    def candidate_edges(self, samples):
        """The (neighbour, new node) pairs the reference loop would hand to ``connect`` for this sample sequence.
        -> vertex states (V,d), edge list (M,2) of node indices (vertices first, then goals), per-edge distances."""
        p = self._params
        d = self._space.dimension
        verts = [np.asarray(self._start, dtype=np.float64)]
        targets = []                                   # per iteration: node index of new_node
        goal_hits = []                                 # (iteration position, goal index, vertex count at that time)
        self._node_counts = []                         # graph nodes (vertices + goals) when each iteration queried its neighbours
        for s in samples:
            for gi, g in enumerate(self._goals):
                if self._space.distance(s, g) < p.goal_tolerance:
                    goal_hits.append((gi, len(verts)))
                    targets.append(-1 - gi)
                    break
            else:
                verts.append(np.asarray(s, dtype=np.float64))
                targets.append(len(verts) - 1)
            self._node_counts.append(len(verts) + len(self._goals))
        V = np.vstack(verts).reshape(-1, d)
        nv = V.shape[0]
        nbr = knn_prefix(V, p.k_nearest)
        src, dst, cnt = [], [], []
        for it, t in enumerate(targets):
            if t >= 0:
                row = nbr[t]
                row = row[row >= 0]
                src.append(row)
                dst.append(np.full(row.shape, t, dtype=np.int64))
                cnt.append(np.full(row.shape, self._node_counts[it], dtype=np.int64))
        # a sample that landed on a goal: k nearest vertices of the goal state among those present at that time
        x32 = V.astype(np.float32)
        for gi, count in goal_hits:
            g32 = np.asarray(self._goals[gi], dtype=np.float32)
            dd = ((x32[:count] - g32) ** 2).sum(axis=1)
            row = np.argsort(dd, kind="stable")[:p.k_nearest]
            src.append(row.astype(np.int64))
            dst.append(np.full(row.shape, nv + gi, dtype=np.int64))
            cnt.append(np.full(row.shape, count + len(self._goals), dtype=np.int64))
        src = np.concatenate(src) if src else np.zeros((0,), dtype=np.int64)
        dst = np.concatenate(dst) if dst else np.zeros((0,), dtype=np.int64)
        cnt = np.concatenate(cnt) if cnt else np.zeros((0,), dtype=np.int64)
        nodes = np.vstack([V] + [np.asarray(g, dtype=np.float64)[None] for g in self._goals])
        keep = self._neighbour_filter(nodes[src], nodes[dst], cnt)
        src, dst = src[keep], dst[keep]
        dist = np.asarray(self._space.distance_batch(nodes[src], nodes[dst]), dtype=np.float64).reshape(-1)
        return V, nodes, np.stack([src, dst], axis=1), dist

    def _neighbour_filter(self, near, state, node_count):
        """Which of the k nearest are kept (graph.py:170: ``norm(q_near - state) < radius``); PRM: radius = inf."""
        return np.ones((near.shape[0],), dtype=bool)

    def plan(self, samples=None):
        if self._start is None:
            raise ValueError("Must set start state before planning")
        if len(self._goals) == 0:
            raise ValueError("Must set goal states before planning")
        if samples is None:
            samples = [self.sample_state() for _ in range(self._params.max_iters)]
        V, nodes, cand, dist = self.candidate_edges(samples)
        self.n_candidate_edges = int(cand.shape[0])
        # zero-length edges (a vertex and itself) are what connect() refuses before looking at the scene
        ok = np.zeros((cand.shape[0],), dtype=bool)
        live = dist > np.finfo(np.float32).eps
        if live.any():
            ok[live] = np.asarray(self._connector.connect_batch(nodes[cand[live, 0]], nodes[cand[live, 1]], dist[live]))
        self.states, self._nodes = V, nodes
        self.edges, self.weights = cand[ok], dist[ok]
        return self

    def solution(self):
        """Shortest start -> goal path over the roadmap (graph.py:199-233), as Nodes; None when no goal is reached."""
        from scipy.sparse import coo_matrix
        from scipy.sparse.csgraph import dijkstra
        if self.edges is None or self.edges.shape[0] == 0:
            return None
        n = self._nodes.shape[0]
        nv = self.states.shape[0]
        # parallel edges (a pair proposed from both ends) collapse to one: COO construction would add their weights
        key = np.sort(self.edges, axis=1)
        _, first = np.unique(key, axis=0, return_index=True)
        e, w = self.edges[first], self.weights[first]
        g = coo_matrix((w, (e[:, 0], e[:, 1])), shape=(n, n)).tocsr()
        dist, pred = dijkstra(g, directed=False, indices=0, return_predecessors=True)
        goals = np.arange(nv, n)
        best = goals[np.argmin(dist[goals])]
        if not np.isfinite(dist[best]):
            return None
        path = [int(best)]
        while path[-1] != 0:
            path.append(int(pred[path[-1]]))
        path.reverse()
        vid = getattr(self, "vertex_ids", None)                      # planners that drop vertices keep upstream's ids
        name = lambda i: f"v_{i if vid is None else vid[i]}" if i < nv else f"g_{i - nv}"      # noqa: E731
        return [Node(id=name(i), state=self._nodes[i], cost=float(dist[i])) for i in path]


class PRMStar(PRM):
    """prm_star.py:14-58: PRM whose neighbours must also lie within the shrinking connection radius
    ``gamma (log n / n)^(1/d)``, n = graph nodes when the sample is inserted."""

    def connection_radius(self, n_nodes):
        from scipy.special import gamma as _gamma
        dim = float(self._space.dimension)
        k = np.asarray(n_nodes, dtype=np.float64)
        V_ball = (np.pi ** (dim / 2.0)) / _gamma((dim / 2.0) + 1.0)
        V_space = self._space.volume
        g = 2.0 * (1.0 + (1.0 / dim)) ** (1.0 / dim) * (V_ball / V_space) ** (1.0 / dim)
        return g * (np.log(k) / k) ** (1 / dim)

    def _neighbour_filter(self, near, state, node_count):
        return np.linalg.norm(near - state, axis=1) < self.connection_radius(node_count)


class RRT(PRM):
    """rrt.py:13-52: the tree grows one steer at a time (each ``steer`` is one device launch over the whole edge);
    nothing to batch across iterations, provided for completeness of the planner family."""

    def plan(self, samples=None):
        if self._start is None:
            raise ValueError("Must set start state before planning")
        if len(self._goals) == 0:
            raise ValueError("Must set goal states before planning")
        p = self._params
        verts = [np.asarray(self._start, dtype=np.float64)]
        edges, weights = [], []
        n_goal_edges = 0
        it = iter(samples) if samples is not None else None
        for _ in range(p.max_iters):
            rand_state = next(it) if it is not None else self.sample_state()
            X = np.asarray(verts, dtype=np.float32)
            d = ((X - np.asarray(rand_state, dtype=np.float32)) ** 2).sum(axis=1)
            near = int(np.argmin(d))
            new_state = self._connector.steer(verts[near], rand_state, distance_func=self._space.distance)
            if new_state is None:
                continue
            for gi, g in enumerate(self._goals):
                if self._space.distance(new_state, g) < p.goal_tolerance:
                    edges.append((near, -1 - gi))
                    weights.append(self._space.distance(new_state, g))
                    n_goal_edges += 1
                    break
            else:
                verts.append(np.asarray(new_state, dtype=np.float64))
                edges.append((near, len(verts) - 1))
                weights.append(self._space.distance(verts[near], new_state))
                continue
            break
        V = np.vstack(verts)
        nv = V.shape[0]
        self.states = V
        self._nodes = np.vstack([V] + [np.asarray(g, dtype=np.float64)[None] for g in self._goals])
        e = np.array([(a, b if b >= 0 else nv + (-1 - b)) for a, b in edges], dtype=np.int64).reshape(-1, 2)
        self.edges, self.weights = e, np.asarray(weights, dtype=np.float64)
        self.n_candidate_edges = len(edges)
        return self


class RRTStar(PRM):
    """rrt_star.py:13-87: steer towards the sample, choose the cheapest parent among the near vertices, rewire the
    others through the new vertex.  One iteration needs up to ``k_nearest`` ``connect`` calls
    (rrt_star.py:51-58); they go to ONE ``connect_batch`` call here (one edge per wavefront), the steer before
    them is one more launch.  The tree (parent, edge weight, cost-to-come per vertex) lives in arrays instead of a
    networkx DiGraph; costs follow ``PlanningGraph.update_costs_recursive`` (graph.py:190-196): parent cost + edge
    weight, pushed down the subtree on every change.

    As upstream: the loop runs all ``max_iters`` iterations (it does not stop at the first goal hit); a steered
    state within ``goal_tolerance`` of a goal is dropped again and its best parent is linked to the goal
    (rrt_star.py:61-70); the connection radius is ``rewire_factor * (log d / d)^(1/n)``, n = graph nodes, goals
    included (rrt_star.py:22-25).  Vertex ids stay stable when a vertex is dropped (upstream's faiss labels become
    positional after ``remove_points`` rebuilds the index, nearest_neighbors.py:49-64 -- evidently unintended)."""

    def connection_radius(self, n_nodes):
        dim = float(self._space.dimension)
        return self._params.rewire_factor * (np.log(dim) / dim) ** (1 / float(n_nodes))

    def _push_costs(self, v):
        stack = [v]
        while stack:
            u = stack.pop()
            for c in self._children[u]:
                self.cost[c] = self.cost[u] + self._wpar[c]
                stack.append(c)

    def plan(self, samples=None):
        if self._start is None:
            raise ValueError("Must set start state before planning")
        if len(self._goals) == 0:
            raise ValueError("Must set goal states before planning")
        p = self._params
        dist = self._space.distance
        dim = self._space.dimension
        cap = p.max_iters + 1
        S = np.zeros((cap, dim), dtype=np.float64)
        S32 = np.zeros((cap, dim), dtype=np.float32)
        alive = np.zeros((cap,), dtype=bool)
        S[0] = self._start
        S32[0] = S[0]
        alive[0] = True
        n = 1
        self.parent = [-1]
        self._wpar = [0.0]
        self.cost = [0.0]
        self._children = [[]]
        self.goal_edges = [dict() for _ in self._goals]            # per goal: parent vertex -> edge weight
        self.n_candidate_edges = 0
        self.n_rewired = 0
        it = iter(samples) if samples is not None else None

        def l2_32(x):
            d = np.zeros((n,), dtype=np.float32)
            x32 = np.asarray(x, dtype=np.float32)
            for c in range(dim):
                t = S32[:n, c] - x32[c]
                d += t * t
            d[~alive[:n]] = np.inf
            return d

        for _ in range(p.max_iters):
            rand_state = next(it) if it is not None else self.sample_state()
            near = int(np.argmin(l2_32(rand_state)))
            new_state = self._connector.steer(S[near], rand_state, distance_func=dist)
            if new_state is None:
                continue
            new = n
            S[new] = new_state
            S32[new] = S[new]
            alive[new] = True
            n += 1
            self.parent.append(-1)
            self._wpar.append(0.0)
            self.cost.append(np.inf)
            self._children.append([])
            radius = self.connection_radius(int(alive[:n].sum()) + len(self._goals))
            d32 = l2_32(S[new])
            order = np.argsort(d32, kind="stable")[:p.k_nearest]
            order = order[np.isfinite(d32[order])]
            nb = order[np.linalg.norm(S[order] - S[new], axis=1) < radius]

            w = np.array([dist(S[j], S[new]) for j in nb], dtype=np.float64)
            ok = np.zeros((nb.shape[0],), dtype=bool)
            live = w > np.finfo(np.float32).eps
            if live.any():
                self.n_candidate_edges += int(live.sum())
                ok[live] = np.asarray(self._connector.connect_batch(S[nb[live]], np.tile(S[new], (int(live.sum()), 1)), w[live]))
            best, best_cost = near, self.cost[near] + dist(S[near], S[new])
            for j, wj, okj in zip(nb, w, ok):
                if okj and self.cost[j] + wj < best_cost:
                    best, best_cost = int(j), self.cost[j] + wj

            for gi, g in enumerate(self._goals):
                if dist(S[new], g) < p.goal_tolerance:
                    alive[new] = False
                    self.goal_edges[gi][best] = dist(S[new], g)
                    break
            else:
                self.parent[new] = best
                self._wpar[new] = dist(S[best], S[new])
                self._children[best].append(new)
                self.cost[new] = self.cost[best] + self._wpar[new]
                for j, okj in zip(nb, ok):
                    j = int(j)
                    if not okj:
                        continue
                    wj = dist(S[new], S[j])
                    if self.cost[new] + wj < self.cost[j]:                  # graph.py:181-187
                        if self.parent[j] < 0:
                            raise ValueError("Nodes should have one parent, found 0")
                        self._children[self.parent[j]].remove(j)
                        self.parent[j], self._wpar[j] = new, wj
                        self._children[new].append(j)
                        self.cost[j] = self.cost[new] + wj
                        self._push_costs(j)
                        self.n_rewired += 1

        keep = np.nonzero(alive[:n])[0]
        self.vertex_ids = keep                                       # upstream ids ``v_<id>`` of the rows of ``states``
        remap = -np.ones((n,), dtype=np.int64)
        remap[keep] = np.arange(keep.shape[0])
        self.states = S[keep]
        nv = keep.shape[0]
        self._nodes = np.vstack([self.states] + [np.asarray(g, dtype=np.float64)[None] for g in self._goals])
        e = [(remap[self.parent[j]], remap[j]) for j in keep if self.parent[j] >= 0]
        wts = [self._wpar[j] for j in keep if self.parent[j] >= 0]
        for gi, ge in enumerate(self.goal_edges):
            for par, wg in ge.items():
                e.append((remap[par], nv + gi))
                wts.append(wg)
        self.edges = np.asarray(e, dtype=np.int64).reshape(-1, 2)
        self.weights = np.asarray(wts, dtype=np.float64)
        return self
