classdef KLTTracker
    % GPU-backed drop-in for modules/KLTTracker.m: same constructor options and track() signature;
    % the body forwards to the vo_klt_mex gateway (libvo_hip.so) instead of vision.PointTracker.
    % Put this directory on the path AFTER setup_env.m so that it shadows the reference file.
    properties
        opts
    end
    methods
        function self = KLTTracker(nv)
            arguments
                nv.NumPyramidLevels double = 4
                nv.MaxBidirectionalError double = 2
                nv.BlockSize double = [41 41]
                nv.MaxIterations double = 50
            end
            self.opts = nv;
        end
        function [points, validity, scores] = track(self, img1, img2, img1_pts)
            % empty input: the gateway returns double.empty(0,2), [], double.empty(0,1)
            [points, validity, scores] = vo_klt_mex(img1, img2, img1_pts, self.opts);
        end
    end
end
